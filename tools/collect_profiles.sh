#!/bin/bash
# Collects the rocprofv3 evidence of profiles/ on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <tag>        -> gpurun_out/profiles_<tag>/
# For the default workload: one --kernel-trace --stats run of the bench (200 timed steps).  For every workload key: five separate --pmc
# passes (never combined with traces) + tools/profile_summary.py -> valu.json / traffic.json entries, tagged with the profiled
# library's rover_version(); then the un-profiled bench lines.  Raw traces stay in /tmp, only summaries are kept.
set -u
TAG=${1:-rXX}
OUT=gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export ROVER_SCENE_CACHE=/tmp/scene_cache
P=/tmp/prof_$TAG
rm -rf "$P"; mkdir -p "$P"
cp profiles/valu.json profiles/traffic.json "$OUT"/ 2>/dev/null
LIB=$(python3 -c "import sys; sys.path.insert(0, '.'); from isaac_rover_amd import _lib; print(_lib.version())")
echo "profiling $LIB"
# the library must be the one the sources in the tree build (an A/B session leaves an experiment's .so in csrc/): refuse otherwise
python3 -c "import sys; sys.path.insert(0, '.'); from isaac_rover_amd import _lib; sys.exit(0 if _lib.version().endswith('src-' + _lib.source_hash()) else 1)" \
  || { echo "librover_step.so was not built from the sources in the tree: run csrc/build.sh first"; exit 1; }
python3 bench.py --steps 5 --warmup 2 --passes 1 --no-cpu-baseline --no-also > /dev/null 2>&1       # builds the scene cache
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 bench.py --steps 200 --warmup 10 --passes 3 --no-cpu-baseline --no-also > "$OUT/${TAG}_bench_under_rocprof.json" 2> $P/stats.err

pmc_passes() {   # $1 = workload key, $2 = file tag, rest = bench arguments
  local key=$1 ftag=$2; shift 2
  local D=$P/$ftag; mkdir -p $D
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc1 -- python3 bench.py --steps 5 --warmup 2 --passes 1 --no-cpu-baseline --no-also "$@" > /dev/null 2> $D/pmc1.err
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D/pmc2 -- python3 bench.py --steps 5 --warmup 2 --passes 1 --no-cpu-baseline --no-also "$@" > /dev/null 2> $D/pmc2.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $D/pmc3 -- python3 bench.py --steps 5 --warmup 2 --passes 1 --no-cpu-baseline --no-also "$@" > /dev/null 2> $D/pmc3.err
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR --output-format csv -d $D/pmc4 -- python3 bench.py --steps 5 --warmup 2 --passes 1 --no-cpu-baseline --no-also "$@" > /dev/null 2> $D/pmc4.err
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $D/pmc5 -- python3 bench.py --steps 5 --warmup 2 --passes 1 --no-cpu-baseline --no-also "$@" > /dev/null 2> $D/pmc5.err
  local extra=""; [ "$ftag" = "$TAG" ] && extra="$P/stats"
  python3 tools/profile_summary.py $extra $D/pmc1 $D/pmc2 $D/pmc3 $D/pmc4 $D/pmc5 --out "$OUT" --tag "$ftag" --workload-key "$key" --lib "$LIB" --merge-into "$OUT"
  tail -c 300 $D/*.err | grep -i "error\|fatal" | head -5
}
pmc_passes E65536_P37_K200_C600 "$TAG"
pmc_passes E4096_P37_K200_C600 "${TAG}_4096envs" --envs-per-gpu 4096
pmc_passes E32768_P37_K200_C600 "${TAG}_32768envs" --envs-per-gpu 32768
pmc_passes E65536_P120_K200_C600 "${TAG}_cfg5" --rays 120 --validate-goals
pmc_passes E65536_P37_K200_C600_fp16_as_shipped "${TAG}_fp16" --ray-precision fp16_as_shipped
pmc_passes E65536_P37_K200_C600_irregular "${TAG}_irregular" --mesh irregular
# un-profiled bench lines (valu.json / traffic.json of this run are in place: the lines carry their roofline fractions)
cp "$OUT"/valu.json "$OUT"/traffic.json profiles/
# (bench.py prints the compact record line on stdout and writes everything it measured to gpurun_out/bench_full.json: both are kept)
b() { local name=$1; shift; python3 bench.py "$@" > "$OUT/${name}.json" 2> /dev/null; cp gpurun_out/bench_full.json "$OUT/${name}_full.json" 2> /dev/null; }
b "${TAG}_bench"                                                  # the driver's command: headline + also{} + cpu_baseline
b "${TAG}_bench_4096envs" --envs-per-gpu 4096 --steps 200 --warmup 100 --no-cpu-baseline --no-also
b "${TAG}_bench_32768envs" --envs-per-gpu 32768 --no-cpu-baseline --no-also
b "${TAG}_bench_cfg5_120rays_goalvalidation" --rays 120 --validate-goals --no-cpu-baseline --no-also
b "${TAG}_bench_fp16_as_shipped" --ray-precision fp16_as_shipped --no-cpu-baseline --no-also
b "${TAG}_bench_mesh_shuffled" --mesh shuffled --no-cpu-baseline --no-also
b "${TAG}_bench_mesh_irregular" --mesh irregular --no-cpu-baseline --no-also
b "${TAG}_bench_native_4096envs" --rays native --envs-per-gpu 4096 --no-cpu-baseline --no-also
# the reference's own operating point: numEnvs 512 (cfg/task/Rover.yaml:11) x its native 1 634 + 26 rays, on the decimated-style mesh
b "${TAG}_bench_native_512envs_irregular" --rays native --envs-per-gpu 512 --mesh irregular --steps 200 --warmup 100 --no-cpu-baseline --no-also
b "${TAG}_bench_native_512envs_irregular_fp16" --rays native --envs-per-gpu 512 --mesh irregular --ray-precision fp16_as_shipped --steps 200 --warmup 100 --no-cpu-baseline --no-also
ls -la "$OUT"
