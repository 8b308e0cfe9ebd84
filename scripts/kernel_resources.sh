#!/bin/bash
# Register / scratch / LDS use of every kernel of one source file, from the gfx950 assembly's metadata (no GPU needed):
#   scripts/kernel_resources.sh gemm_tn.hip [name filter (regex on the demangled name)]
cd "$(dirname "$0")/../conformer-pytorch-lightning_amd/csrc" || exit 1
out=/tmp/$(basename "$1").s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S "$1" -o "$out" -Wno-pass-failed 2>/dev/null || exit 1
awk '/^ +\.group_segment_fixed_size:/ {l=$2} /^ +\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_spill_count:/ {vs=$2} /\.sgpr_spill_count:/ {ss=$2} /\.private_segment_fixed_size:/ {p=$2} /\.agpr_count:/ {a=$NF} /\.wavefront_size:/ {printf "%s vgpr %s agpr %s sgpr %s spill v%s s%s scratch %s lds %s\n", name, v, a, s, vs, ss, p, l}' "$out" \
  | c++filt | sed -e 's/(anonymous namespace):://g' -e 's/void //' | grep -E "${2:-.}"
