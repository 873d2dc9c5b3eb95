#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4prof; mkdir -p $O
b4() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --requests-per-gpu 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('batch4 $tag', round(d['ms_per_step'],4), d['lossless_fraction'])"; }
python -m pytest tests/test_hip_batch.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1; do echo "== DFL_RING_F32=$v"; DFL_RING_F32=$v SRC=frag python scripts/bench_gemm_batch.py downf of 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
for rep in 1 2; do
  b4 "A=2 ring_f32=0" DFL_RING_VARIANT=4 DFL_RING_F32=0
  b4 "A=3" DFL_RING_VARIANT=0
  b4 "A=2" DFL_RING_VARIANT=4
  b4 "A=4" DFL_RING_VARIANT=2
  b4 "A=2 attn_wgs=512" DFL_RING_VARIANT=4 DFL_ATTN_HEAD_WGS_MULTI=512
  b4 "A=2 attn_tiles=4" DFL_RING_VARIANT=4 DFL_ATTN_HEAD_TILES=4
  b4 "A=2 attn_pair=0" DFL_RING_VARIANT=4 DFL_ATTN_HEAD_PAIR=0
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_b4
DFL_RING_VARIANT=4 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_b4 -- python3 $R/bench.py --steps 24 --warmup 2 --no-cpu-baseline --requests-per-gpu 4 --eager > $O/bench_b4_profiled.json 2> $O/trace_b4.err
python3 $R/scripts/kstats_trace.py $O/trace_b4 3000 | tee $O/b4_kernels_by_grid_A2.txt | head -16
rm -rf $O/trace_b4
