for ev in "ROVER_RAYCAST_VARIANT=3" "ROVER_RAYCAST_VARIANT=4 ROVER_LANE_ROCKS=0" "ROVER_RAYCAST_VARIANT=4 ROVER_LANE_ROCKS=1"; do
  echo -n "fp16 65536 $ev: "
  env $ev python bench.py --ray-precision fp16_as_shipped --passes 5 --no-torch-ref --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['kernel'])"
done
