# Developer tool: rebuild the libraries of earlier rounds that tools/ab_step.py compares with the
# current one (same hipcc flags as the package build), from this repository's own history:
#   bash tools/build_ab_libs.sh            -> tools/ab/librbg_r2.so (b580a39), tools/ab/librbg_r3.so (f1ff6be)
#   bash tools/build_ab_libs.sh r4b:4ee8681 r4c:ed60d0f    any other commit, as tag:rev
# The .so files are git-ignored; they travel to the GPU box with the snapshot.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/ab"
FL="--offload-arch=gfx950 -O3 -fPIC -shared -ffp-contract=off -std=c++17 -mllvm -amdgpu-sched-strategy=max-ilp"
for spec in ${@:-r2:b580a39 r3:f1ff6be}; do
  tag=${spec%%:*}; rev=${spec##*:}
  tmp=$(mktemp -d)
  git -C "$ROOT" archive "$rev" routing_board_generation_amd/csrc include | tar -x -C "$tmp"
  (cd "$tmp" && ${HIPCC:-/opt/rocm/bin/hipcc} $FL -o "$ROOT/tools/ab/librbg_$tag.so" routing_board_generation_amd/csrc/prw_kernels.hip)
  rm -rf "$tmp"
  echo "built tools/ab/librbg_$tag.so from $rev"
done
