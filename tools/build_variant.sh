#!/bin/bash
# Build a copy of the library for an A/B measurement: build_variants/libptrace_<name>.so (git-ignored, travels with gpurun;
# load it with PTRACE_LIB=build_variants/libptrace_<name>.so).
#   tools/build_variant.sh <name> [-DMACRO=value ...]          the working tree's sources with extra compiler flags
#   tools/build_variant.sh <name> --rev <commit> [-D...]       the sources of a commit (git worktree under /tmp)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
SRC=$ROOT
if [ "$1" = "--rev" ]; then
  REV=$2; shift 2
  SRC=$(mktemp -d /tmp/ptvariant.XXXXXX)
  git -C $ROOT archive $REV pytracer_amd/csrc include | tar -x -C $SRC
fi
mkdir -p $ROOT/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    -Wno-pass-failed -cuid=libptrace "$@" -o $ROOT/build_variants/libptrace_$NAME.so $SRC/pytracer_amd/csrc/ptrace.hip
[ "$SRC" != "$ROOT" ] && rm -rf $SRC
echo "built build_variants/libptrace_$NAME.so"
