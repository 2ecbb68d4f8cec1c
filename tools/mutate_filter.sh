#!/bin/bash
# Mutation check of the scattered-ray filter's tests: build copies of the library with the filter made WRONG in ways a
# careless edit could, then run tests/test_gpu_probes.py (filtered against exhaustive query) and the random scenes on each.
# Every mutant must fail.  usage (repository root; the second half on the GPU box): tools/mutate_filter.sh build | run
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build_variants
mkdir -p $OUT
mutate() {  # name, sed expression applied to pt_query.h / ptrace.hip
  local name=$1 file=$2 expr=$3
  local tmp=$(mktemp -d)
  mkdir -p $tmp/pytracer_amd $tmp/include   # (the sources include ../../include/ptrace.h: keep that shape)
  cp -r $ROOT/pytracer_amd/csrc $tmp/pytracer_amd/csrc
  cp $ROOT/include/*.h $tmp/include/
  sed -i "$expr" $tmp/pytracer_amd/csrc/$file
  if cmp -s $tmp/pytracer_amd/csrc/$file $ROOT/pytracer_amd/csrc/$file; then echo "mutant $name: the edit did not apply"; exit 1; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -fPIC -shared -Wno-unused-function \
      -Wno-pass-failed -o $OUT/libptrace_fmut_$name.so $tmp/pytracer_amd/csrc/ptrace.hip
  rm -rf $tmp
  echo "built $name"
}
if [ "$1" = "build" ]; then
  # 1: radii 2 % too small in the tables           2: the margin for the rounding of o dropped AND the slack inverted
  # 3: shadow rays clamped to half the segment      4: balls behind the origin "seen" through the wrong sign of the clamp
  mutate r98 ptrace.hip 's|(double)\*r \* (double)\*r \* (1.0 + 8.1e-6)|(double)*r * (double)*r * 0.96|'
  mutate slack pt_query.h 's|__builtin_amdgcn_rsqf(dd) \* 1.0000041f|__builtin_amdgcn_rsqf(dd) * 0.9995f|'
  mutate halfseg pt_query.h 's|const float tlen = ANYHIT ? (float)tmax \* (dd \* rn) \* (1.0f + 1e-5f) : 0.0f;|const float tlen = ANYHIT ? 0.5f * (float)tmax * (dd * rn) : 0.0f;|'
  mutate behind pt_query.h 's|__builtin_fmaxf(vd.x, 0.0f), __builtin_fmaxf(vd.y, 0.0f)|__builtin_fminf(vd.x, 0.0f), __builtin_fminf(vd.y, 0.0f)|'
  exit 0
fi
cd $ROOT
for m in r98 slack halfseg behind; do
  echo "== mutant $m"
  PTRACE_LIB=$OUT/libptrace_fmut_$m.so timeout -k 10 300 python -m pytest tests/test_gpu_probes.py -q -x -k "filtered_query" 2>&1 | tail -1
  PTRACE_LIB=$OUT/libptrace_fmut_$m.so PT_FUZZ_SEEDS=40 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "random_scenes or c4 or c3_pathtracer" 2>&1 | tail -1
done
echo "== the library as shipped"
timeout -k 10 300 python -m pytest tests/test_gpu_probes.py -q -x -k "filtered_query" 2>&1 | tail -1
