#!/bin/bash
# Collects the rocprofv3 evidence of profiles/ on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <tag>        -> gpurun_out/profiles_<tag>/
# One --kernel-trace --stats run of the default bench (200 timed steps), four separate --pmc passes (never combined
# with traces), un-profiled bench lines for configs[1], [2], [4]; raw traces stay in /tmp, only summaries are kept.
set -u
TAG=${1:-rXX}
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
P=/tmp/prof_$TAG
rm -rf "$P"; mkdir -p "$P"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_under_rocprof.json" 2> $P/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $P/pmc1.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/pmc2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $P/pmc2.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $P/pmc3 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $P/pmc3.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR --output-format csv -d $P/pmc4 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $P/pmc4.err
python3 tools/profile_summary.py $P/stats $P/pmc1 $P/pmc2 $P/pmc3 $P/pmc4 --out "$OUT" --tag "$TAG" --workload-key E65536_P37_K200_C600
python3 bench.py > "$OUT/${TAG}_bench.json" 2> /dev/null
python3 bench.py --envs-per-gpu 4096 --steps 1000 --warmup 100 > "$OUT/${TAG}_bench_4096envs.json" 2> /dev/null
python3 bench.py --rays 120 --validate-goals > "$OUT/${TAG}_bench_cfg5_120rays_goalvalidation.json" 2> /dev/null
python3 bench.py --ray-precision fp16_as_shipped --no-cpu-baseline > "$OUT/${TAG}_bench_fp16_as_shipped.json" 2> /dev/null
tail -c 600 $P/*.err | tail -20
ls -la "$OUT"
