#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds oracle/_ref/libref.so from the reference sources where they lie
# under /root/reference (nothing is copied into this repository; oracle/_ref/ is git-ignored).
#
# The reference as written is MSVC-only.  Three constructs stop g++ (all checked with g++ 11.4):
#   1. src/utility/numeric.hpp:10-11  `typedef IndexType IndexType;` re-declares a template parameter;
#   2. src/utility/vectorised.hpp:31,36,103,108,130-137,142,147  `.m128_f32[i]`, `.m256_f32[i]`,
#      `.m256i_u32[i]` are members of MSVC's __m128/__m256/__m256i unions;
#   3. CMakeLists.txt:23 omits -mfma although _mm_fmadd_ps is used (src/utility/vectorised.hpp:407).
# The recipe therefore preprocesses the harness (which #includes the reference headers in place),
# rewrites exactly those constructs in the preprocessed stream, and compiles the stream.  No other
# text of the reference is touched, and no header, library or tool is substituted.
#
# -ffp-contract=off is mandatory: with contraction the hit point `origin + t*direction`
# (src/geometry.hpp:30-32) is fused and the in-leaf-box test (src/bsp.hpp:80-84) flips for the floor
# of the built-in scene, which lies exactly on the root box face.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${RT_REFERENCE_ROOT:-/root/reference}"
out="$here/_ref"
if [ ! -d "$ref/src" ]; then
    echo "build_ref.sh: $ref/src not present, skipping (expected on the GPU box)" >&2
    exit 0
fi
mkdir -p "$out"
flags=(-std=c++17 -O3 -mavx2 -mfma -ffp-contract=off -DNDEBUG -fopenmp -fPIC)
g++ "${flags[@]}" -E -I"$ref/src" -I"$ref/libraries/glm" "$here/ref_harness.cpp" \
  | sed -e 's/typedef IndexType IndexType;//' \
        -e 's/typedef SizeType SizeType;//' \
        -e 's/\(VertexRange\|TriRange\)::\(IndexType\|SizeType\)/std::uint32_t/g' \
        -e 's/data\.m128_f32\[/((float*)\&data)[/g' \
        -e 's/data\.m256_f32\[/((float*)\&data)[/g' \
        -e 's/data\.m256i_u32\[/((std::uint32_t*)\&data)[/g' \
  | g++ "${flags[@]}" -x c++-cpp-output - -shared -o "$out/libref.so"
echo "built $out/libref.so"

# The same harness with contraction allowed, as the author's MSVC /fp:fast build had it: only used to show where the
# reference's own example_render.png got its darker floor from (tests/test_oracle_vs_ref.py); never a source of expected
# values.
fast=("${flags[@]/-ffp-contract=off/-ffp-contract=fast}")
g++ "${fast[@]}" -E -I"$ref/src" -I"$ref/libraries/glm" "$here/ref_harness.cpp" \
  | sed -e 's/typedef IndexType IndexType;//' \
        -e 's/typedef SizeType SizeType;//' \
        -e 's/\(VertexRange\|TriRange\)::\(IndexType\|SizeType\)/std::uint32_t/g' \
        -e 's/data\.m128_f32\[/((float*)\&data)[/g' \
        -e 's/data\.m256_f32\[/((float*)\&data)[/g' \
        -e 's/data\.m256i_u32\[/((std::uint32_t*)\&data)[/g' \
  | g++ "${fast[@]}" -x c++-cpp-output - -shared -o "$out/libref_contracted.so"
echo "built $out/libref_contracted.so"

# The reference program with render() bound to librt_amd.so (oracle/ref_dropin.cpp): same recipe, linked
# against the product library, which it finds next to the repository's package at run time.
lib="$here/../raytracing_amd"
if [ -f "$lib/librt_amd.so" ]; then
    g++ "${flags[@]}" -E -I"$ref/src" -I"$ref/libraries/glm" -I"$here/../include" "$here/ref_dropin.cpp" \
      | sed -e 's/typedef IndexType IndexType;//' \
            -e 's/typedef SizeType SizeType;//' \
            -e 's/\(VertexRange\|TriRange\)::\(IndexType\|SizeType\)/std::uint32_t/g' \
            -e 's/data\.m128_f32\[/((float*)\&data)[/g' \
            -e 's/data\.m256_f32\[/((float*)\&data)[/g' \
            -e 's/data\.m256i_u32\[/((std::uint32_t*)\&data)[/g' \
      | g++ "${flags[@]}" -x c++-cpp-output - -o "$out/ref_main_gpu" -L"$lib" -lrt_amd \
            -Wl,-rpath,'$ORIGIN/../../raytracing_amd' -Wl,-rpath-link,/opt/rocm/lib
    echo "built $out/ref_main_gpu"
fi
