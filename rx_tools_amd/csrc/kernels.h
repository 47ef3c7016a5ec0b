/* kernels.h -- internal C interface between the C host code (rxgpu_*.c) and the HIP
 * kernels (fm_kernels.hip, power_kernels.hip).  Launchers only enqueue on the given
 * stream and return the hipError_t of the launch as int (0 == hipSuccess).
 * Not part of the public ABI (include/rxgpu.h is). */
#ifndef RXGPU_KERNELS_H
#define RXGPU_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the snapshot of a $RXGPU_* tuning knob (rxgpu_rt.c: read at rxgpu_init, object creation and rxgpu_knobs_reload -- never on a launch
 * path); NULL when unset */
const char *rxgpu_knob(const char *name);

/* complex samples one workgroup of the fast boxcar decimator covers (power of two) */
#define RXK_DEC_SPAN 16384
#define RXK_DEC_SPAN_LOG2 14
/* largest boxcar the fast decimator takes: every workgroup but the last must hold an
 * output boundary */
#define RXK_DEC_MAX_DS (RXK_DEC_SPAN / 2)

/* device-resident scalars of one rx_fm run: carries in, carries out, status */
typedef struct rxk_fm_dev {
	/* carry in (host fills before the run) */
	int in_now_r, in_now_j, in_prev_index;
	int in_pre_r, in_pre_j;
	int in_deemph_avg;
	int in_now_lpr, in_prev_lpr_index;
	int in_dc_avg;
	/* carry out (kernels fill) */
	int out_now_r, out_now_j, out_prev_index;
	int out_pre_r, out_pre_j;
	int out_deemph_avg;
	int out_now_lpr, out_prev_lpr_index;
	int out_dc_avg;
	/* status */
	int flag_cnt;          /* libm-discriminator samples needing host re-evaluation */
	int reserved;
	int err;               /* != 0: device-side invariant violated */
} rxk_fm_dev;

/* which decimated samples belong to which callback block */
typedef struct rxk_fm_blocks {
	int first_mode;                /* RXK_FIRST_LOWPASS: block b owns lp[(b*n+p0)/ds .. ((b+1)*n+p0)/ds); UNIFORM: lp[b*k .. (b+1)*k) */
	int ds, p0;
	unsigned long long n, k, n_blocks;
	int post;                      /* > 1: the ranges are asked in samples after low_pass_simple (-o) */
} rxk_fm_blocks;

#define RXK_FLAG_CAP 4096

/* a libm-discriminator sample the device could not decide: its index in the run's demodulated stream and the two
 * decimated samples it is made of (a = lp[m], b = lp[m-1] or the carried pre_r/pre_j) -- all the host needs to
 * re-evaluate it with its own libm */
typedef struct rxk_flag_rec {
	unsigned long long m;
	int ar, aj, br, bj;
} rxk_flag_rec;

/* |v - rint(v)| below this flags a libm sample, v = atan2(cj,cr) / 3.14159 * 16384 (rtl_fm.c:476-483).  Device atan2
 * (OCML, <= 2 ulp) and glibc's (<= 1 ulp) differ by <= 3 ulp(pi) = 1.33e-15; the division by 3.14159 is correctly
 * rounded on both sides (<= 1.1e-16 each way on a quotient <= 1.0000009), the multiplication by 2^14 is exact:
 * |v_dev - v_host| <= (1.33e-15 / 3.14159 + 2.2e-16) * 16384 = 1.06e-11 < 2^-36.  2^-33 leaves a factor 8. */
#define RXK_LIBM_WINDOW 0x1p-33

/* what k_fm_block_dd leaves for the host: low_pass's and fm_demod's carries, the libm samples it could not decide */
#define RXK_BLK_FLAGS 64
typedef struct rxk_blk_out {
	int now_r, now_j, prev_index, pre_r, pre_j, flag_cnt, pad[2];
	rxk_flag_rec rec[RXK_BLK_FLAGS];
} rxk_blk_out;
/* low_pass (rtl_fm.c:351-371) + fm_demod (584-615, -A std | fast | ale) of one pre-scaled block in one launch (the drop-in's single blocks);
 * lp, pcm: (p0 + n) / ds entries; audio_in: the three seeds rxk_ch_audio takes ({avg, now_lpr, prev_lpr_index}) */
int rxk_fm_block_dd(void *stream, const int16_t *blk, unsigned n, int ds, int p0, int now_r, int now_j, int pre_r, int pre_j,
                    int custom_atan, uint32_t *lp, uint32_t *lp_host, int16_t *pcm, int16_t *keep, rxk_blk_out *out, int *audio_in, int avg,
                    int now_lpr, int prev_lpr_index);
/* deemph_filter + low_pass_real on one short row held in LDS (W <= RXK_ROW_AUDIO_MAX samples); audio: {avg, now_lpr, prev_lpr_index} in, the same
 * three out at audio_out; row_out: the result in device memory (row itself for in place, NULL for none) */
#define RXK_ROW_AUDIO_MAX 24000u
int rxk_fm_row_audio(void *stream, const int16_t *row, int16_t *row_out, unsigned W, int deemph, int a, int warm, int serial, int fast, int slow,
                     unsigned J, const int *audio, int *audio_out, int16_t *row_h, int *audio_h, const void *hdr, void *hdr_h, unsigned hdr_words);
/* row_h, audio_h, hdr_h: device-addressable HOST memory that receives the result row, the three audio carries and a copy of hdr[0 .. hdr_words) */

enum { RXK_FIRST_LOWPASS = 0, RXK_FIRST_UNIFORM = 1 };

/* F0+F1+F2 fused (rtl_fm.c:845-848, 309-327, 351-371): cs16 -> scaled -> rotated ->
 * boxcar sums.  T complex samples (T % 4 == 0), 4 <= ds <= RXK_DEC_MAX_DS.
 * lp_raw[m] valid for outputs completed strictly inside one workgroup span, head/tail
 * hold the per-workgroup partial sums at the seams (packed int16 I | Q<<16, mod 2^16). */
/* pcm != NULL: also the -A fast discriminator for every output but the first two of each span.
 * lp_sparse != 0 (with pcm, ds <= RXK_LP_SPARSE_MAX_DS): lowpassed[] is only an intermediate then, and only the
 * entries rxk_fm_disc(sparse) will read are stored (the second and the last output of every span). */
#define RXK_FFT_XROW 17                  /* dwords an LDS transpose area reserves per thread-row (fft_device.h: fft_geom<M>::XROW) */
#define RXK_LP_SPARSE_MAX_DS 512
int rxk_fm_decimate(void *stream, const int16_t *iq, unsigned long long T, int ds, int p0,
                    int prescaled, int rotate, uint32_t *lp_raw, uint32_t *head, uint32_t *tail, int lp_sparse, int16_t *pcm, int pcm_chl2);
/* ... on the raw capture of an -E rdc run whose blocks are whole spans: the block averages (rxk_fm_rdc with out == NULL: sums + recursion only) are
 * subtracted from the scaled samples inside the decimator */
int rxk_fm_disc_rdc(void *stream, const int16_t *iq, unsigned long long T, int ds, int p0, unsigned long long n_per_block, int rotate, int seams, const uint32_t *lp_raw,
                    const uint32_t *head, const uint32_t *tail, uint32_t *lp, unsigned long long M, int custom_atan, int16_t *pcm, rxk_fm_dev *dev,
                    rxk_flag_rec *flag_list, int *flag_cnt, int sparse, unsigned long long n_blocks, const int *atan_lut, int lp_sparse, int flag_all, int pcm_chl2,
                    const int *rdc_avg);      /* rxk_fm_disc behind rxk_fm_decimate_rdc */
int rxk_fm_decimate_rdc(void *stream, const int16_t *iq, unsigned long long T, int ds, int p0, int rotate, uint32_t *lp_raw, uint32_t *head, uint32_t *tail,
                        int lp_sparse, int16_t *pcm, int pcm_chl2, const int *rdc_avg, unsigned spans_per_block);
/* pcm_chl2 != 0: pcm[] is written in the tiled layout of the lane-per-chunk audio kernels (chunks of 2^pcm_chl2 samples, see
 * pcm_index in fm_kernels.hip); rxk_fm_disc takes the same argument */

/* the direct form for small decimation (RXK_DEC_SMALL_MIN <= ds <= RXK_DEC_SMALL_MAX, raw cs16 input, T % 4 == 0): scaled samples staged
 * in LDS once, one thread per output, no seams, -A fast discriminator for every output into pcm[] (linear or tiled); lowpassed[] is not
 * kept.  rxk_fm_disc(sparse, seams = 2) then redoes the run's first two outputs and every block's first one and takes the carries. */
#define RXK_DEC_SMALL_MIN 4
#define RXK_DEC_SMALL_MAX 32
int rxk_fm_decimate_small(void *stream, const int16_t *iq, unsigned long long T, int ds, int p0, int rotate, unsigned long long M,
                          int16_t *pcm, int pcm_chl2);

/* same maths, one thread per output, any ds >= 1 and any block length; writes final lp[] */
int rxk_fm_decimate_generic(void *stream, const int16_t *iq, unsigned long long T, int ds, int p0,
                            unsigned long long n_per_block, int prescaled, int rotate,
                            const rxk_fm_dev *dev, uint32_t *lp, unsigned long long M);

/* seam fix-up + F5/F6 discriminator (rtl_fm.c:584-615, 476-513) over M decimated samples,
 * plus the exact int32 tail sums for the low_pass carry.  seams != 0: lp_raw/head/tail come
 * from rxk_fm_decimate and lp[] is produced; seams == 0: lp[] is already final. */
int rxk_fm_disc(void *stream, const int16_t *iq, unsigned long long T, int ds, int p0,
                unsigned long long n_per_block, int prescaled, int rotate, int seams,
                const uint32_t *lp_raw, const uint32_t *head, const uint32_t *tail,
                uint32_t *lp, unsigned long long M, int first_mode, unsigned long long uniform_k,
                int custom_atan, int do_tail, int16_t *pcm, rxk_fm_dev *dev, rxk_flag_rec *flag_list, int *flag_cnt,
                int sparse, unsigned long long n_blocks, const int *atan_lut, int lp_sparse, int flag_all, int pcm_chl2);
/* flag_all: report EVERY libm sample as undecided (test hook: the host then re-evaluates all of them); 2: and store a
 * deliberately wrong value for it, so that only the host fix-up + the redo of the audio stages can produce the right output */
/* lp_sparse: lp_raw[] holds only what rxk_fm_decimate(lp_sparse) stored; any other window this kernel needs (a
 * block's first output and its predecessor, the last two outputs) is summed again from iq and stored into lp[] */
/* pcm == NULL: only finish lp[] (+ the low_pass carry); the discriminator runs later on the final lp[] */
/* sparse != 0 (after rxk_fm_decimate with pcm): only the two seam outputs of every span, each block's
 * first (libm) output and the last output are processed */

/* F8 de-emphasis (rtl_fm.c:667-682) as a tree scan over chunk maps (see fm_kernels.hip).
 * group = 16 or 64 candidate lanes; RXK_DEEMPH_FAN tables compose into one per level. */
#define RXK_DEEMPH_FAN 16
/* per workgroup of RXK_DEEMPH_WG_CHUNKS consecutive chunks: the composite of their tables = level 0 of the tree
 * (p_tab, p_lo, p_gap) and, per chunk, its start state for every candidate of the workgroup's first chunk (pre,
 * `group` ints per chunk: rxk_fm_deemph_apply picks one); chunk = 2^k >= max(128, warm), warm % 8 == 0 */
#define RXK_DEEMPH_WG_CHUNKS 64
int rxk_fm_deemph_scan(void *stream, const int16_t *pcm, unsigned long long M, int a, int group,
                       int chunk, int warm, int lo0, int hi0, int *pre, int *p_tab, int *p_lo, int *p_gap,
                       rxk_fm_dev *dev);
int rxk_fm_deemph_up(void *stream, unsigned long long n_child, int group, const int *tab, const int *lo,
                     const int *gap, int *p_tab, int *p_lo, int *p_gap);
int rxk_fm_deemph_top(void *stream, int n, int group, const int *tab, const int *lo, const int *gap,
                      int *start, rxk_fm_dev *dev);
int rxk_fm_deemph_down(void *stream, unsigned long long n_child, int group, const int *tab, const int *lo,
                       const int *p_start, int *start);
/* p_start: the exact start state of every workgroup of chunks (level 0 of the tree, walked down) */
int rxk_fm_deemph_apply(void *stream, const int16_t *pcm, unsigned long long M, int a, int group, int chunk,
                        const int *pre, const int *p_lo, const int *p_start, int16_t *y);
/* The same two stages on the TILED stream, for the small-decimation chains (fm_kernels.hip, "F8 + F9 on the tiled stream"):
 * lane per chunk straight from HBM, compact 16-byte chunk tables (ctab), the tree's first level on those (up0 / down0, the
 * levels above are rxk_fm_deemph_up/_top/_down), then replay + low_pass_real inline -> out[].  rxk_fm_deemph_tiled_ok
 * returns 0 or the chunk log2 (7/8) to hand to the decimator as pcm_chl2. */
int rxk_fm_deemph_tiled_ok(int a, int group, int chunk, int fast, int slow);
int rxk_fm_deemph_scan_t(void *stream, const int16_t *pcm_t, unsigned long long M, int a, int group, int chl2, int warm, int lo0, int gap_w,
                         void *ctab, rxk_fm_dev *dev);
int rxk_fm_deemph_up0(void *stream, unsigned long long n_chunks, int group, const void *ctab, int *p_tab, int *p_lo, int *p_gap);
int rxk_fm_deemph_down0(void *stream, unsigned long long n_chunks, const void *ctab, const int *p_start, int *start);
int rxk_fm_deemph_apply_rs_t(void *stream, const int16_t *pcm_t, unsigned long long M, int a, int chl2, const int *start, int fast, int slow,
                             int16_t *out, rxk_fm_dev *dev);
/* any a, any state: one lane, serial (degenerate fallback, still on the device) */
int rxk_fm_deemph_serial(void *stream, const int16_t *pcm, unsigned long long M, int a, int16_t *y, rxk_fm_dev *dev);

/* F9 low_pass_real (rtl_fm.c:389-409): J outputs from n inputs, closed-form windows */
int rxk_fm_resample(void *stream, const int16_t *y, unsigned long long n, int fast, int slow,
                    unsigned long long J, int16_t *out, rxk_fm_dev *dev);
/* power squelch (rtl_fm.c:781-790, rms 739-757) per callback block: below[b] = rms < level, and such blocks are zeroed */
int rxk_fm_squelch(void *stream, uint32_t *lp, rxk_fm_blocks blk, int level, int *below, int *sr_out);   /* sr_out[b]: the block's rms, rtl_fm.c:781 */
/* am/usb/lsb_demod (rtl_fm.c:617-656) on the final decimated IQ */
int rxk_fm_simple_demod(void *stream, const uint32_t *lp, unsigned long long M, int mode, int output_scale, int16_t *pcm);
/* dc_block_audio_filter (rtl_fm.c:684-697): per-block mean, first-order recursion across blocks, subtract */
int rxk_fm_dc_block(void *stream, int16_t *y, unsigned long long M, rxk_fm_blocks blk, int adc_block_const,
                    long long *sums, int *avgs, rxk_fm_dev *dev);
/* chained runs: carries-out -> carries-in on the device (advance != 0); either way the audio-stage carries-in of the
 * run (deemph_avg, now_lpr, prev_lpr_index, dc_avg) are copied to snap[0..4) so that the stages can be redone later */
/* a few bytes as one wave (not a blit kernel): device -> device, or device -> pinned host */
int rxk_copy_small(void *stream, void *dst, const void *src, unsigned bytes);
/* a device buffer into a page-locked host mirror (both 16-byte aligned), 16-byte units: the drop-in's lowpassed[] on its way home */
int rxk_copy_mirror(void *stream, void *dst, const void *src, unsigned bytes);
int rxk_fm_carry_advance(void *stream, rxk_fm_dev *dev, int advance, int *snap);
/* redo of audio stages: their carries-in from a snapshot (snap != NULL) or from the carries-out of the run before */
int rxk_fm_audio_carry(void *stream, rxk_fm_dev *dev, const int *snap);
/* carries when a stage is disabled */
int rxk_fm_passthrough_carry(void *stream, rxk_fm_dev *dev, int deemph_off, int resample_off);

/* F3 fifth_order cascade (rtl_fm.c:411-440, 764-769): one pass over every block.
 * in/out are per-block contiguous arrays of packed IQ (uint32) except pass 0, which reads
 * the raw cs16 stream and applies scale + rotate on the fly.
 * n_in complex samples per block in, n_out = ceil(n_in/2) out; hist_i/hist_q: the carried
 * lp_i_hist[pass]/lp_q_hist[pass] (6 int16 each) for block 0; hist_out receives the new ones. */
int rxk_fm_fifth_pass(void *stream, const void *in, int in_is_raw, int prescaled, int rotate,
                      unsigned long long n_blocks, unsigned n_in, unsigned in_stride, uint32_t *out,
                      unsigned out_stride, const int16_t *hist_in, int16_t *hist_out);
/* `fuse` (1..3) passes of the cascade in one launch; n % RXK_FIFTH_TILE == 0.  Raw input (stage2 == 0) also takes fuse == 4 and 5 (n % 2^fuse == 0;
 * seams then holds 5 * fuse dwords per block): those run in registers (k_fm_fifth_regn: lane-contiguous runs, neighbours by DPP, no LDS
 * tiles, no barriers; $RXGPU_FR_GENERIC=1: fuse == 3 too), everything else in the LDS-tiled kernel.
 * stage2 == 0: raw cs16 input (scale + rotate on the fly; packed-int16 arithmetic is exact through three
 * passes from raw).  stage2 != 0: packed level samples in, the reference's int arithmetic; hist_in/hist_out
 * point at the first pass of the group.  seams: n_blocks * 5 * fuse dwords of scratch. */
#define RXK_FIFTH_TILE 2048
int rxk_fm_fifth_seams(void *stream, const void *in, int stage2, int rotate, unsigned long long n_blocks, unsigned n, int fuse,
                       const int16_t *hist_in, int16_t *hist_out, uint32_t *seams);
int rxk_fm_fifth_fused(void *stream, const void *in, int stage2, int rotate, unsigned long long n_blocks, unsigned n, int fuse,
                       const int16_t *hist_in, int16_t *hist_out, uint32_t *seams, uint32_t *out);
/* A three-pass cascade on raw input with everything behind it in the same launch (k_fm_fifth_regn<.., 3, DD>): droop FIR (fir = the
 * cic_9_tables row, HOST pointer, or NULL) and the -A fast discriminator; pcm linear or tiled (pcm_chl2).  seams: rxk_fm_fifth_seams with
 * fuse = 3; tails (10 dwords per block) and the new droop history: rxk_fm_fifth_tails; every block's first sample and pre_r/pre_j out:
 * rxk_fm_dd_edges on `edges` (2 dwords per block), which alone touches dev and the flag list. */
int rxk_fm_fifth_dd(void *stream, const void *in, int rotate, unsigned long long n_blocks, unsigned n, int fuse, const uint32_t *seams,
                    const uint32_t *tails, const int *fir, int16_t *pcm, int pcm_chl2, uint32_t *edges);
int rxk_fm_fifth_tails(void *stream, const void *in, int rotate, unsigned long long n_blocks, unsigned n, int fuse, const int16_t *droop_in,
                       int16_t *droop_out, uint32_t *tails);
int rxk_fm_dd_edges(void *stream, const uint32_t *edges, unsigned long long n_blocks, unsigned long long K, int16_t *pcm, int pcm_chl2,
                    rxk_fm_dev *dev, rxk_flag_rec *flag_list, int *flag_cnt, int flag_all);
/* F12 generic_fir droop compensation (rtl_fm.c:442-465, 771-776) over the concatenated stream */
int rxk_fm_droop(void *stream, const uint32_t *in, unsigned long long M, const int *fir,
                 const int16_t *hist_in, int16_t *hist_out, uint32_t *out);

/* F12 + fm_demod (-A fast; each block's first sample through libm) in one pass over the post-cascade stream; uniform blocks of
 * uniform_k samples.  lp_out (optional): the FIR output; pcm: linear or tiled (pcm_chl2). */
int rxk_fm_droop_disc(void *stream, const uint32_t *in, unsigned long long M, const int *fir, const int16_t *hist_in, int16_t *hist_out,
                      uint32_t *lp_out, unsigned long long uniform_k, int16_t *pcm, int pcm_chl2, rxk_fm_dev *dev, rxk_flag_rec *flag_list,
                      int *flag_cnt, int flag_all);

/* The literal per-block path: full_demod's -F cascade and what follows it on ONE block's int16 lowpassed[] of L int16, indexed exactly
 * like the C loops (rtl_fm.c:411-465, 584-665, 739-757, 764-790), for blocks whose sample count is not a multiple of 2^passes
 * (lp_len >> i turns odd, I and Q yield different counts).  hist: I then Q (6 + 6 int16; the droop FIR's 9 + 9). */
int rxk_fm_fifth_lit(void *stream, const int16_t *in, int16_t *out, int L, const int16_t *hist_in, int16_t *hist_out);
int rxk_fm_droop_lit(void *stream, const int16_t *in, int16_t *out, int L, const int *fir, const int16_t *hist_in, int16_t *hist_out);
int rxk_fm_squelch_lit(void *stream, int16_t *lp, int L, int level, int *below, int *sr_out);
enum { RXK_LIT_FM = 0, RXK_LIT_AM = 1, RXK_LIT_USB = 2, RXK_LIT_LSB = 3, RXK_LIT_RAW = 4 };     /* == RXGPU_MODE_* */
/* results to pcm[m0 ..]: L/2 of them (raw: L); fm: the block's first sample through libm against dev->in_pre (pre_from_out: out_pre, a
 * later block of the run), then dev->out_pre = lp[L-2], lp[L-1] when L >= 2 */
int rxk_fm_demod_lit(void *stream, const int16_t *lp, int L, int mode, int custom_atan, int output_scale, int16_t *pcm,
                     unsigned long long m0, rxk_fm_dev *dev, int pre_from_out, rxk_flag_rec *flag_list, int *flag_cnt,
                     const int *atan_lut, int flag_all);

/* rtlsdr_callback's scale + rotate alone (rtl_fm.c:845-857), n_complex samples */
/* ... with the raw block and a second copy of the result in page-locked HOST memory the device addresses (zero-copy: one launch, no DMA) */
int rxk_fm_prestage_zc(void *stream, const int16_t *in_host, unsigned n_complex, int rotate, int16_t *out_dev, int16_t *out_host);
int rxk_fm_prestage(void *stream, const int16_t *in, unsigned n_complex, int rotate, int16_t *out);

/* channeliser (extension): fix_fft per window, selected bins as [channel][window]; then fm_demod per channel */
/* fused != 0 (allowed when rxk_ch_fused_ok): the FFT kernel also demodulates (-A fast) every window but the first of each
 * group and keeps only the bins rxk_ch_demod(sparse) needs in chan_lp; out / out_stride / pre_out as for rxk_ch_demod */
int rxk_ch_fused_ok(int bin_e, unsigned long long wpb, int custom_atan, int n_channels);   /* 0, or the group size to pass as `fused` / `sparse` */
int rxk_ch_fft(void *stream, const int16_t *iq, unsigned long long total_windows, int bin_e, const uint32_t *twiddle,
               int first_bin, int n_channels, uint32_t *chan_lp, int fused, int16_t *out, unsigned long long out_stride, int *pre_out);
/* the NCO mode of the channeliser: chan_lp[c][w] = low_pass at downsample N of the scaled capture mixed by channel c's NCO; tw_full: N packed (cos, sin) */
int rxk_ch_nco(void *stream, const int16_t *iq, unsigned long long total_windows, int bin_e, const uint32_t *tw_full, int first_bin, int n_channels,
               uint32_t *chan_lp);
int rxk_ch_demod(void *stream, const uint32_t *chan_lp, unsigned long long total_windows, unsigned long long wpb, int n_channels,
                 int custom_atan, const int *pre_in, int *pre_out, int16_t *out, unsigned long long out_stride,
                 rxk_fm_dev *dev, unsigned long long *flag_list, int sparse);

/* per-channel deemph_filter + low_pass_real on the channeliser's [channel][window] output (one workgroup per channel);
 * audio_in/out: {avg, now_lpr, prev_lpr_index} per channel; y_rows: scratch rows when slow > 0.  warm: samples that bring any two
 * int16 start states within 64 of each other (host: deemph_warm); serial != 0: one thread per channel does the recursion */
/* the same stages on a (segment, channel) grid (deemph on, a in 2..64, carried states inside int16), reading the demodulated rows from one buffer
 * and writing the audio to another: rxk_ch_audio_seg_ok says whether rows of W samples can go that way (else rxk_ch_audio serves them in place);
 * rxk_ch_audio_chunks: chunk tables per channel.  ctab: n_channels * that many 16-byte tables, seg_start: as many ints (every chunk's start state) */
unsigned rxk_ch_audio_chunks(unsigned long long W, int warm, unsigned *chunk_out);
int rxk_ch_audio_seg_ok(unsigned long long W, int warm, int fast, int slow);
int rxk_ch_audio_seg(void *stream, const int16_t *in_rows, unsigned long long in_stride, int16_t *out_rows, unsigned long long out_stride,
                     unsigned long long W, int n_channels, int a, int warm, int fast, int slow, const int *audio_in, int *audio_out,
                     void *ctab, int *seg_start);
int rxk_ch_audio(void *stream, int16_t *rows, unsigned long long row_stride, unsigned long long W, int n_channels, int deemph, int a,
                 int warm, int serial, int fast, int slow, unsigned long long J, const int *audio_in, int *audio_out,
                 int16_t *y_rows, unsigned long long y_stride);

/* ------------------------------------------------------------- rx_power */

/* P1,P4-P8 fused (rtl_power.c:715-720, 744-770) for ds == 1 or pre-downsampled input:
 * one workgroup per (tune, pass-group): remove_dc over the tune buffer, then per FFT block
 * window -> fix_fft in LDS -> |X|^2, accumulated into d_avg[tune][n] (atomic add / max).
 * eff_len = int16 per tune actually transformed (buf_len / ds). */
int rxk_pw_fft(void *stream, const int16_t *in, size_t tune_stride, size_t pass_stride, int passes,
               int tunes, int bin_e, int eff_len, int dc_len, const int *window, const uint32_t *twiddle,
               int peak_hold, int passes_per_group, long long *avg, long long *partial, size_t partial_cap);
/* partial (optional, partial_cap int64): with ceil(passes / passes_per_group) * tunes * max(1, 4096 >> bin_e) << bin_e of them the register-
 * blocked kernels (bin_e 8..13) write per-group spectra there and one reduction adds them to avg -- for sweeps of few tunes, where
 * every pass lands on the same bins and atomics on avg[] dominate */
/* samples[t] += blocks_per_tune * ds * passes (rtl_power.c:769) */
int rxk_pw_samples(void *stream, int *samples, int tunes, int add);
/* the drop-in's P1 (rtl_power.c:715-720): rows[i] = device-visible address of tune i's page-locked buf16, row_bytes a multiple of 16;
 * one launch copies every row into out[rows][row_bytes] */
int rxk_pw_gather_rows(void *stream, const void *const *d_rows, int rows, size_t row_bytes, int16_t *out);
/* host row r (a tune's avg[], page-locked, device-visible address d_rows[r]) += or max= acc row r of row_bytes / 8 int64, acc zeroed */
int rxk_pw_merge_rows(void *stream, void *const *d_rows, int rows, size_t row_bytes, long long *acc, int peak_hold, int host_rows_zero);   /* host_rows_zero: the rows hold zeros -- written, not read */
/* P2 boxcar (rtl_power.c:723-733): every buffer of buf_len int16 -> same-size buffer whose
 * complex slot k holds the wrapped sum of samples [k*ds,(k+1)*ds), zero elsewhere */
int rxk_pw_boxcar(void *stream, const int16_t *in, int16_t *out, size_t n_bufs, int buf_len, int ds, int n_write);
/* P2 through rxk_fm_decimate (prescaled, no rotation) when buffers hold whole windows: finish the per-span seam entries */
int rxk_pw_boxcar_seams(void *stream, uint32_t *lp, const uint32_t *head, const uint32_t *tail, unsigned long long T, int ds,
                        long long *dc_sums, const int *wave_sums, unsigned spans_per_buf);
/* dc_sums != NULL: the spans' wave sums (rxk_pw_boxcar_sums) and seam outputs join their buffers' remove_dc sums, spans_per_buf = buf_len / 2 / RXK_DEC_SPAN */
/* the boxcar over buffers that are whole spans, every wave leaving the sums of the outputs it stored in wave_sums[(span * 4 + wave) * 2 + {0: I, 1: Q}] */
int rxk_pw_boxcar_sums(void *stream, const int16_t *iq, unsigned long long T, int ds, uint32_t *lp_raw, uint32_t *head, uint32_t *tail, int *wave_sums);
/* n_write: complex slots per buffer actually produced (what the transform will read); <= 0: all of them */
/* P3 one stateless fifth_order pass on I and Q (rtl_power.c:582-607, 656-662): n_in complex
 * samples per buffer -> ceil(n_in/2); strides in complex samples */
int rxk_pw_fifth(void *stream, const int16_t *in, int16_t *out, size_t n_bufs, int n_in, int in_stride, int out_stride);
/* `fuse` (1..3) of those passes in one LDS-tiled launch (the rx_fm cascade kernel with ease-in instead of carried history);
 * n % RXK_FIFTH_TILE == 0 */
int rxk_pw_fifth_fused(void *stream, const int16_t *in, unsigned long long n_bufs, unsigned n, unsigned in_stride, int fuse,
                       int16_t *out, unsigned out_stride);
/* four stateless fifth_order passes (+ droop FIR when fir_host != NULL: cic_9_tables[4], fir_dev the same table on the device) in registers, one launch +
 * a fix-up of each buffer's first samples; sums != NULL: remove_dc's sums of every output buffer into sums[2 b], sums[2 b + 1] (zeroed by the caller) */
int rxk_pw_fifth_regn(void *stream, const int16_t *in, unsigned long long n_bufs, unsigned n, unsigned in_stride, int lv, const int *fir_dev, const int *fir_host,
                      int16_t *out, unsigned out_stride, long long *sums, int *wave_part);
/* lv = 2, 3 or 4 stateless passes in registers.  sums != NULL: wave_part = rxk_pw_fifth_regn_parts(n_bufs, n, lv) int PAIRS of scratch (every wave leaves its share
 * there, the fix kernel adds them: sums need no zeroing) */
unsigned long long rxk_pw_fifth_regn_parts(unsigned long long n_bufs, unsigned n, int lv);
/* generic_fir (rtl_power.c:626-654) over n complex samples per buffer */
int rxk_pw_droop(void *stream, const int16_t *in, int16_t *out, size_t n_bufs, int n, int stride, const int *fir);
/* P9 rms_power sums: t[b] = sum s, p[b] = sum s^2 (int64, exact) per buffer, then the fp64
 * dc correction + accumulate (rtl_power.c:418-428) */
int rxk_pw_rms_sums(void *stream, const int16_t *in, size_t n_bufs, int buf_len, long long *t, long long *p);
int rxk_pw_rms_apply(void *stream, const long long *t, const long long *p, int passes, int tunes, int buf_len,
                     int peak_hold, long long *avg, int *samples);

/* -E rdc, dc_block_raw_filter (rtl_fm.c:699-721, 850-852): per block sums -> one-thread recursion over the blocks
 * (state[2] = dc_avgI, dc_avgQ, read and written on the device) -> out = rotate16_90(scaled - avg) as int16 pairs.
 * sums: 2 int64 per block, avg: 2 int per block (workspace). */
int rxk_fm_rdc(void *stream, const int16_t *iq, unsigned long long n_blocks, unsigned long long n_per_block, int prescaled,
               int rotate, int rdc_block_const, int *state, long long *sums, int *avg, int16_t *out);
/* -o, low_pass_simple (rtl_fm.c:373-387): out[j] = (int16) sum of in[j*step .. j*step+step) */
int rxk_fm_post_downsample(void *stream, const int16_t *in, unsigned long long n_out, int step, int16_t *out);

/* fix_fft for 2^15 < N <= 2^21 (rtl_power.c:485): the same network on a scratch copy in HBM, one launch per stage.
 * scratch: cap_blocks * 2^bin_e dwords; dc: 2 ints per (pass, tune) */
/* bin_e 14 .. 21, eff_len a multiple of 2^(bin_e+1): the register-blocked transform as two to four launches (radix-16 passes through a
 * scratch copy in HBM until a sub-transform fits one workgroup, then the independent sub-transforms), same scratch / dc workspaces as
 * rxk_pw_fft_big; returns < 0 if the geometry does not fit */
/* where the dc workspace (2 ints per (pass, tune), then -- 16-byte aligned -- 2 int64 per (pass, tune)) keeps remove_dc's sums */
long long *rxk_pw_dc_sums(int *dc, size_t n_pass_tunes);
int rxk_pw_fft_mid(void *stream, const int16_t *in, size_t tune_stride, size_t pass_stride, int passes, int tunes,
                   int bin_e, int eff_len, const int *window, const uint32_t *twiddle, int peak_hold,
                   uint32_t *scratch, size_t cap_blocks, int *dc, long long *avg, long long *partial, size_t partial_cap, int dc_sums_done,
                   int *samples, int samples_add);     /* samples != NULL: samples[tune] += samples_add in one of the launches (no rxk_pw_samples needed) */
/* workgroups the second launch aims for; partial needs (RXK_PWM_TARGET_WG / 4 + tunes * blocks per tune) * 2^bin_e int64 to be used */
#ifndef RXK_PWM_TARGET_WG
#define RXK_PWM_TARGET_WG 2048
#endif
int rxk_pw_fft_big(void *stream, const int16_t *in, size_t tune_stride, size_t pass_stride, int passes, int tunes,
                   int bin_e, int eff_len, const int *window, const uint32_t *twiddle, int peak_hold,
                   uint32_t *scratch, size_t cap_blocks, int *dc, long long *avg);

/* ---- rx_sdr output converters (sdr_kernels.hip), rtl_sdr.c:354-391; n16 = int16 count, device pointers */
int rxk_sdr_cs16_to_8(void *stream, const int16_t *in, unsigned long long n16, int is_unsigned, uint8_t *out);
int rxk_sdr_cs16_to_cf32(void *stream, const int16_t *in, unsigned long long n16, float *out);
int rxk_sdr_cs12_to_cs16(void *stream, const uint8_t *in, unsigned long long n_elems, int16_t *out);

/* diagnostics: the converters' loop with the arithmetic taken out (sdr_kernels.hip k_diag_stream); units of 16 B read (8 B for mode 2) */
int rxk_diag_stream(void *stream, int mode, const void *in, unsigned long long units, void *out);

#ifdef __cplusplus
}
#endif
#endif
