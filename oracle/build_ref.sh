#!/usr/bin/env bash
# oracle/build_ref.sh -- build oracle/_ref/libmsdr_ref.so from the reference's OWN sources,
# compiled where they lie under /root/reference.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).
#
# What is built (every compiled line of arithmetic is the reference's text, none is ours):
#   arm_fir_fast_q15   src/CMSIS_5/arm_fir_fast_q15.c:60-329   (A4)
#   arm_fir_init_q15   src/CMSIS_5/arm_fir_init_q15.c:78-138   (A3, Cortex-M4 branch)
#   arm_copy_q15       src/CMSIS_5/arm_copy_q15.c:48-98
#   arm_sqrt_q31       src/CMSIS_5/arm_sqrt_q31.c:50-138       (A5, Teensy-3.2 AM variant)
#   arm_sqrt_f32       src/CMSIS_5/arm_math.h:5733-5758        (A5, Teensy-3.6 AM variant: a static inline of the header, through a caller)
#   calc_FIR_coeffs / m_sinc / Izero   Minimal-SDR.ino:782-899 (A9)
#   row f4 (spectrum, UI.cpp:520-592 -> arm_rfft_q15): arm_cfft_radix4_q15.c (arm_cfft_radix4_q15,
#   arm_radix4_butterfly_q15), arm_cfft_radix4_init_q15.c, arm_bitreversal.c (arm_bitreversal_q15),
#   arm_split_rfft_q15 (arm_rfft_q15.c:128-283), arm_rfft_init_q15.c, arm_const_structs.c,
#   arm_common_tables.c (twiddle / bit-reversal / realCoef tables as data)
#   rows A7 / f1, the arithmetic primitives only: signed_multiply_32x16b/t, signed_saturate_rshift, pack_16b_16b / 16t_16t / 16t_16b
#   (src/Audio/utility/dspinst.h:34-51, :70-92, :151-184, the header's own plain-C `KINETISL` bodies)
#
# How the two non-trivial cases are handled WITHOUT hand-written stand-ins:
#  * arm_fir_fast_q15.c needs __SMLAD/__SMLADX/__PKHBT.  The reference header arm_math.h
#    carries generic-C definitions of all three (arm_math.h:492-501, :895-918) but hides them
#    behind `#if !defined(ARM_MATH_DSP)`, and hard-wires ARM_MATH_CM4 -> ARM_MATH_DSP (:294,:328).
#    We derive _ref/gen/arm_math.h from the reference header by dropping that ONE `#define
#    ARM_MATH_DSP` line (awk below, diff printed), and feed the .c through stdin so that its
#    `#include "arm_math.h"` resolves to the derived copy.  The FIR source itself has no
#    ARM_MATH_DSP conditionals, so the algorithm compiled is the Cortex-M4 one.
#  * calc_FIR_coeffs lives inside the Arduino sketch.  Its three functions are cut out of the
#    .ino by name anchors and emitted callee-first (the Arduino IDE would have generated the
#    prototypes) into _ref/gen/fir_design_extract.cpp behind the derived arm_math.h
#    (float32_t, PI; the CM4 branch of the header does not parse as C++ off-target).  On a Teensy `PI` comes from the un-vendored Arduino.h (a double
#    literal); arm_math.h:365-367 only supplies a float fallback.  Both variants are built:
#    calc_FIR_coeffs (float PI, what the vendored sources alone give) and
#    calc_FIR_coeffs_pid (-DPI=<Arduino's double literal>).
#
#  * The FFT sources DO carry `#if defined (ARM_MATH_DSP)` branches, and the Teensy 3.6 runs the DSP
#    ones.  They are compiled behind a three-line wrapper fed through stdin: include the derived
#    header (so the reference's generic-C __SHADD16/__QADD16/__SMUAD/__SMUSDX/... are defined),
#    `#define ARM_MATH_DSP`, include the source.  arm_rfft_q15() itself calls arm_cfft_q15(), whose
#    bit reversal arm_bitreversal_16 exists only as ARM assembly (arm_bitreversal2.S) -- so those
#    two functions are NOT built; arm_split_rfft_q15 is cut out of arm_rfft_q15.c by name anchors,
#    and the 64-point complex FFT is pinned through the reference's all-C entry point
#    arm_cfft_radix4_q15 (same butterfly function, C bit reversal arm_bitreversal_q15).
#
# NOT buildable here, by rule (would need stand-ins for the un-vendored Teensyduino core
# Arduino.h/AudioStream.h and for ARM inline asm): filter_biquad.cpp, freq_conv.cpp,
# demodulation() as a whole.  Those rows are "parity unpinned" -- see DESIGN.md.
set -euo pipefail
REF=${MSDR_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
GEN="$OUT/gen"
if [ ! -d "$REF/src/CMSIS_5" ]; then
  echo "build_ref.sh: $REF not present -- keeping prebuilt $OUT (if any)"; exit 0
fi
mkdir -p "$GEN"
C5="$REF/src/CMSIS_5"
CFLAGS="-O2 -w -fPIC -fno-strict-aliasing -fwrapv -ffp-contract=off"

# (1) derived header: select the reference's own generic-C intrinsics
awk '/#elif defined \(ARM_MATH_CM4\)/{f=1} f&&/#define ARM_MATH_DSP/{print "  /* build_ref.sh: ARM_MATH_DSP left undefined here */"; f=0; next} {print}' \
    "$C5/arm_math.h" > "$GEN/arm_math.h"
echo "--- derived arm_math.h differs from the reference header by:"
diff "$C5/arm_math.h" "$GEN/arm_math.h" || true

# (2) the FIR, through stdin so "arm_math.h" resolves to the derived copy first
gcc $CFLAGS -x c -I"$GEN" -I"$C5" -c -o "$OUT/arm_fir_fast_q15.o" - < "$C5/arm_fir_fast_q15.c"
# (3) the rest compile unmodified against the unmodified header
for f in arm_fir_init_q15 arm_copy_q15 arm_sqrt_q31; do
  gcc $CFLAGS -c -o "$OUT/$f.o" "$C5/$f.c"
done

# (4) FIR designer, cut from the sketch by function-name anchors
INO="$REF/Minimal-SDR.ino"
{
  echo '// GENERATED by oracle/build_ref.sh from the reference sketch -- never committed.'
  echo '#include <math.h>'
  echo '#include <stdint.h>'
  echo '#include "arm_math.h"   /* the derived copy (generic-C branch): the CM4 branch does not parse as C++ off-target */' 
  grep -E '^#define (PIH|TPI) ' "$INO"
  echo 'extern "C" {'
  awk '/^float32_t Izero/{p=1} p{print} /^} \/\/ END Izero/{p=0}' "$INO"
  awk '/^float m_sinc/{p=1} p{print} p&&/^}/{p=0}' "$INO"
  awk '/^void calc_FIR_coeffs/{p=1} p{print} /^} \/\/ END calc_FIR_coeffs/{p=0}' "$INO"
  echo '}'
} > "$GEN/fir_design_extract.cpp"
g++ $CFLAGS -fpermissive -I"$GEN" -I"$C5" -c -o "$OUT/fir_design.o" "$GEN/fir_design_extract.cpp"
g++ $CFLAGS -fpermissive -I"$GEN" -I"$C5" -DPI=3.1415926535897932384626433832795 \
    -Dcalc_FIR_coeffs=calc_FIR_coeffs_pid -Dm_sinc=m_sinc_pid -DIzero=Izero_pid \
    -c -o "$OUT/fir_design_pid.o" "$GEN/fir_design_extract.cpp"

# (5) row f4: the q15 FFT pieces, DSP branches selected (see header comment)
printf '#include "arm_math.h"\n#define ARM_MATH_DSP\n#include "%s"\n' "$C5/arm_cfft_radix4_q15.c" \
  | gcc $CFLAGS -x c -I"$GEN" -I"$C5" -c -o "$OUT/arm_cfft_radix4_q15.o" -
{
  echo '#include "arm_math.h"'
  echo '#define ARM_MATH_DSP'
  awk '/^}/{body=1} body&&/^void arm_split_rfft_q15\(/{p=1} p{print} p&&/^}/{p=0}' "$C5/arm_rfft_q15.c"   # the definition, not the prototype
} > "$GEN/split_rfft_extract.c"
gcc $CFLAGS -I"$GEN" -I"$C5" -c -o "$OUT/split_rfft.o" "$GEN/split_rfft_extract.c"
for f in arm_cfft_radix4_init_q15 arm_bitreversal arm_rfft_init_q15 arm_rfft_init_q31 arm_const_structs arm_common_tables; do
  gcc $CFLAGS -c -o "$OUT/$f.o" "$C5/$f.c"
done

# (6) the Teensy Audio library's DSP-instruction wrappers (src/Audio/utility/dspinst.h): what AudioFilterBiquad::update
# (filter_biquad.cpp:54-74) and the front end (mixer.cpp, input_adc.cpp) compute with.  The header carries the Cortex-M4 versions as inline
# assembly and, for the Cortex-M0+ of a Teensy LC, the SAME functions in plain C behind `#elif defined(KINETISL)` -- for smulwb / smulwt
# (signed_multiply_32x16b/t), ssat-with-shift (signed_saturate_rshift) and the pkhbt / pkhtb packers.  Those C bodies are compiled here
# as they stand; the only lines of ours are callers with external linkage (the header's functions are `static inline`).  The
# multiply-ACCUMULATE forms the biquad uses (smlawb / smlawt, dspinst.h:235-249) exist as assembly only and are not built: the header's
# own comment defines them as `sum + ((a * b[15:0]) >> 16)`, i.e. the function pinned here plus a 32-bit add.
{
  echo '#include <stdint.h>'
  echo '#define KINETISL'
  echo "#include \"$REF/src/Audio/utility/dspinst.h\""
  echo 'int32_t dspinst_signed_multiply_32x16b(int32_t a, uint32_t b) { return signed_multiply_32x16b(a, b); }'
  echo 'int32_t dspinst_signed_multiply_32x16t(int32_t a, uint32_t b) { return signed_multiply_32x16t(a, b); }'
  echo 'int32_t dspinst_signed_saturate_rshift(int32_t v, int bits, int rshift) { return signed_saturate_rshift(v, bits, rshift); }'
  echo 'uint32_t dspinst_pack_16b_16b(int32_t a, int32_t b) { return pack_16b_16b(a, b); }'
  echo 'uint32_t dspinst_pack_16t_16t(int32_t a, int32_t b) { return pack_16t_16t(a, b); }'
  echo 'uint32_t dspinst_pack_16t_16b(int32_t a, int32_t b) { return pack_16t_16b(a, b); }'
} > "$GEN/dspinst_callers.c"
gcc $CFLAGS -c -o "$OUT/dspinst.o" "$GEN/dspinst_callers.c"
# (7) arm_sqrt_f32 -- the AM / CW branch's square root on a Teensy 3.6 (Minimal-SDR.ino:611-612) -- is a `static inline` of the reference
# header itself (arm_math.h:5733-5758: sqrtf for in >= 0, else 0 and ARM_MATH_ARGUMENT_ERROR): a caller with external linkage, as above.
printf '#include "arm_math.h"\nint cmsis_arm_sqrt_f32(float32_t in, float32_t *pOut) { return (int)arm_sqrt_f32(in, pOut); }\n' \
  | gcc $CFLAGS -x c -I"$GEN" -I"$C5" -c -o "$OUT/arm_sqrt_f32.o" -

gcc -shared -o "$OUT/libmsdr_ref.so" "$OUT"/dspinst.o "$OUT"/arm_sqrt_f32.o "$OUT"/arm_fir_fast_q15.o "$OUT"/arm_fir_init_q15.o \
    "$OUT"/arm_copy_q15.o "$OUT"/arm_sqrt_q31.o "$OUT"/fir_design.o "$OUT"/fir_design_pid.o \
    "$OUT"/arm_cfft_radix4_q15.o "$OUT"/split_rfft.o "$OUT"/arm_cfft_radix4_init_q15.o "$OUT"/arm_bitreversal.o \
    "$OUT"/arm_rfft_init_q15.o "$OUT"/arm_rfft_init_q31.o "$OUT"/arm_const_structs.o "$OUT"/arm_common_tables.o -lm
rm -f "$OUT"/*.o
if nm -D --undefined-only "$OUT/libmsdr_ref.so" | grep -v -e GLIBC -e ' w ' ; then echo "build_ref.sh: unresolved symbols above"; exit 1; fi
echo "built $OUT/libmsdr_ref.so:"
nm -D --defined-only "$OUT/libmsdr_ref.so" | awk '$2 ~ /[TW]/ {print "   " $3}' | grep -v '^   _' || true
# The derived header and the extracted functions are intermediate files made from the reference's text: they served the build and go
# (KEEP_GEN=1 keeps them for inspection).  What stays under _ref/ is the binary only, so nothing of the reference's source -- in
# any form -- rides along when the repository is copied to a GPU box ("nothing from /root/reference travels").
if [ "${KEEP_GEN:-0}" != "1" ]; then rm -rf "$GEN"; fi
