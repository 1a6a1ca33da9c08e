import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
def t(m,k,n):
    x=torch.randn(m,k,device="cuda"); w=torch.randn(n,k,device="cuda"); b=torch.randn(n,device="cuda")
    for _ in range(5): y=torch.nn.functional.leaky_relu(torch.nn.functional.linear(x,w,b))
    torch.cuda.synchronize(); a,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): y=torch.nn.functional.linear(x,w,b)
    e.record(); torch.cuda.synchronize(); lin=a.elapsed_time(e)/20
    a.record()
    for _ in range(20): y=torch.nn.functional.leaky_relu(torch.nn.functional.linear(x,w,b))
    e.record(); torch.cuda.synchronize(); both=a.elapsed_time(e)/20
    print(f"{m}x{k}->{n}: linear {lin*1e3:.1f} us ({2*m*k*n/lin/1e9:.1f} TF), +leakyrelu {both*1e3:.1f} us")
for m,k,n in ((65536,634,80),(65536,1112,80),(65536,80,60),(65536,124,256),(65536,256,160),(65536,160,128),(65536,128,2)):
    t(m,k,n)
