// fm_kernels.hip -- gfx950 kernels for the rx_fm stream pre-stage + full_demod() chain.
//
// Reference behaviour (file:line under /root/reference/src):
//   F0 scale        rtl_fm.c:845-848     F1 rotate16_90   rtl_fm.c:309-327
//   F2 low_pass     rtl_fm.c:351-371     F3 fifth_order   rtl_fm.c:411-440, 764-769
//   F5 fm_demod     rtl_fm.c:584-615     F6 polar_disc_fast/fast_atan2 rtl_fm.c:485-513
//   F8 deemph       rtl_fm.c:667-682     F9 low_pass_real rtl_fm.c:389-409
//   F12 generic_fir rtl_fm.c:442-465
//
// All arithmetic is integer and bit-exact with the C reference except where the reference
// itself goes through libm (polar_discriminant, rtl_fm.c:476-483): there the device
// evaluates in fp64 and flags results too close to a truncation boundary for the host to
// re-evaluate with the same libm the reference uses.
//
// Data layout: the IQ stream is int16 I,Q interleaved (cs16) exactly as SoapySDR delivers
// it; one complex sample = one dword.  A lane reads 4 consecutive samples with one
// global_load_dwordx4 (1 KiB per wave instruction, fully coalesced).  Decimated IQ is kept
// packed (I | Q<<16) mod 2^16, which is exactly the int16 truncation the reference applies
// when it stores int sums back into lowpassed[].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "fft_device.h"

typedef unsigned long long u64;
typedef long long i64;
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DEC_THREADS 256
#define DEC_WAVE_SPAN (RXK_DEC_SPAN / 4)     // samples per wave
#define DEC_TILE 256                         // samples per wave load (64 lanes x 4)
#define DEC_TILES (DEC_WAVE_SPAN / DEC_TILE)
#define DEC_BATCH 8

// The kernels by pipeline stage, one translation unit (the parts share their device helpers and are not compiled alone):
#include "fm_part_helpers.inc"
#include "fm_part_decimate.inc"
#include "fm_part_audio.inc"
#include "fm_part_cascade.inc"
#include "fm_part_chan.inc"
#include "fm_part_launch.inc"
