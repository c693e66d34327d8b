#!/bin/bash
# build_ref.sh -- TEST INFRASTRUCTURE ONLY.
# Compiles the reference's own dependency-free Chamfer lines, from where they
# lie under /root/reference, into oracle/_ref/libref_nndistance.so (git-ignored,
# travels to the GPU box as a binary).  See oracle/ref_nndistance_shim.cpp for
# what is extracted and why the whole TU cannot be built.  Flags follow the
# reference's own recipe (tf_nndistance_compile.sh: g++ -O2) plus
# -ffp-contract=off, which is what its shipped x86-64 object does (no FMA).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${CLOUDAAE_REFERENCE:-/root/reference}"
SRC="$REF/tf_ops/nn_distance/tf_nndistance.cpp"
OUT="$HERE/_ref"
if [ ! -f "$SRC" ]; then
    echo "build_ref.sh: $SRC not present; skipping (prebuilt $OUT is used if it exists)" >&2
    exit 0
fi
# guard the line numbers against a different reference revision
want=fa07b4fb8bcaa8ca14599ae0b4720693
have="$(md5sum "$SRC" | cut -d' ' -f1)"
if [ "$want" != "$have" ]; then
    echo "build_ref.sh: unexpected revision of tf_nndistance.cpp ($have)" >&2
    exit 1
fi
mkdir -p "$OUT"
TMP="$(mktemp -d "$OUT/.extract.XXXXXX")"
trap 'rm -rf "$TMP"' EXIT
sed -n '21,43p' "$SRC" > "$TMP/nnsearch.inc"
sed -n '126,163p' "$SRC" > "$TMP/nngrad.inc"
g++ -std=c++11 -O2 -ffp-contract=off -fPIC -shared -fvisibility=hidden \
    -DREF_NNSEARCH_INC="\"$TMP/nnsearch.inc\"" -DREF_NNGRAD_INC="\"$TMP/nngrad.inc\"" \
    "$HERE/ref_nndistance_shim.cpp" -o "$OUT/libref_nndistance.so"
echo "built $OUT/libref_nndistance.so"

# ---- the reference's GPU kernels for gfx950 (oracle/ref_gpu_shim.hip): tf_sampling_g.cu whole, tf_nndistance_g.cu:5-151 ----
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SAMP="$REF/tf_ops/sampling/tf_sampling_g.cu"
NNDG="$REF/tf_ops/nn_distance/tf_nndistance_g.cu"
if [ ! -x "$HIPCC" ] || [ ! -f "$SAMP" ] || [ ! -f "$NNDG" ]; then
    echo "build_ref.sh: no hipcc or no reference .cu files; libref_gpu.so not built" >&2
    exit 0
fi
if [ "$(md5sum "$SAMP" | cut -d' ' -f1)" != b810331dc78b18bf821c80580faa47d5 ] ||
   [ "$(md5sum "$NNDG" | cut -d' ' -f1)" != 30a89b07b33f9faa7af54b998cd04c8e ]; then
    echo "build_ref.sh: unexpected revision of the reference's .cu files" >&2
    exit 1
fi
sed -n '5,151p' "$NNDG" > "$TMP/nndistance_g.inc"       # the kernels and the forward launcher; see ref_gpu_shim.hip
"$HIPCC" --offload-arch=gfx950 -O2 -ffp-contract=off -fPIC -shared -fvisibility=hidden \
    -DREF_SAMPLING_CU="\"$SAMP\"" -DREF_NNDISTANCE_INC="\"$TMP/nndistance_g.inc\"" \
    "$HERE/ref_gpu_shim.hip" -o "$OUT/libref_gpu.so"
echo "built $OUT/libref_gpu.so"
