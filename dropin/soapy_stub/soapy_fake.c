/* Stand-in libSoapySDR: a capture-replay device.  No radio, no DSP.
 *
 * Implements the handful of SoapySDR C entry points rx_tools links against (see SoapySDR/Device.h next to this file) on top of a
 * cs16 sample source, so that rx_tools' own translation units link and run end to end on a machine without SoapySDR or a radio:
 *   * dropin/Makefile links it into rx_fm / rx_power when no real SoapySDR is found (SOAPY=stub); the capture then comes from a file:
 *       $SOAPY_FAKE_FILE       raw interleaved int16 I,Q (what `rx_sdr -F CS16` writes)
 *       $SOAPY_FAKE_PACE       real-time factor of the replay: 1 = at the sample rate the program set, 0.25 = four times slower,
 *                              unset/0 = as fast as it is read
 *       $SOAPY_FAKE_MAX_READ   elements per readStream at most (a real device returns what it has, not what was asked for)
 *       $SOAPY_FAKE_EOF_SIGINT at the end of the file: raise SIGINT once, after this many milliseconds (what a user's ^C does to
 *                              rx_fm, rtl_fm.c:274-278); the read itself reports SOAPY_SDR_STREAM_ERROR
 *   * oracle/Makefile links it into the reference-built checker objects (TEST INFRASTRUCTURE), which install an in-memory source
 *     and hooks through the soapy_fake_* calls below.
 * readStream hands out consecutive chunks and reports SOAPY_SDR_STREAM_ERROR once the source is exhausted (after calling the
 * optional end-of-stream hook).
 */
#define _GNU_SOURCE
#include <signal.h>
#include <stdio.h>
#include <time.h>
#include <unistd.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <SoapySDR/Device.h>
#include <SoapySDR/Formats.h>

struct SoapySDRDevice { double freq, rate, bw; };
struct SoapySDRStream { int active; };

static struct SoapySDRDevice g_dev;
static struct SoapySDRStream g_stream;

static const int16_t *g_src;      /* interleaved I,Q */
static size_t g_src_len;          /* complex samples */
static size_t g_src_pos;
static size_t g_max_chunk;        /* 0 = hand out whatever is asked for */
static size_t g_elem = 4;         /* bytes per element of the source (cs16 = 4, cs12 = 3) */
static void (*g_eos_hook)(void);
static void (*g_pace_hook)(void); /* called before every data read but the first */
static size_t g_reads;
static const double *g_freq_script;   /* getFrequency answers from this list, round robin (no retune needed) */
static size_t g_freq_n, g_freq_pos;
static const void *g_discard;     /* reads into this buffer are flush reads: zero-fill, consume nothing */

void soapy_fake_set_source(const int16_t *iq, size_t n_complex, size_t max_chunk)
{
	g_src = iq; g_src_len = n_complex; g_src_pos = 0; g_max_chunk = max_chunk;
}
void soapy_fake_set_elem_size(size_t bytes) { g_elem = bytes ? bytes : 4; }
void soapy_fake_set_discard_buffer(const void *p) { g_discard = p; }
void soapy_fake_set_freq_script(const double *f, size_t n) { g_freq_script = f; g_freq_n = n; g_freq_pos = 0; }
void soapy_fake_set_eos_hook(void (*fn)(void)) { g_eos_hook = fn; }
void soapy_fake_set_pace_hook(void (*fn)(void)) { g_pace_hook = fn; g_reads = 0; }
size_t soapy_fake_position(void) { return g_src_pos; }

/* ---- file replay ($SOAPY_FAKE_FILE), set up at the first read */
static int g_env_done;
static double g_pace;             /* real-time factor, 0 = unpaced */
static long g_eos_sigint_ms = -1;
static double g_t0;
static size_t g_paced;            /* elements handed out since g_t0 */

static double mono_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + (double)ts.tv_nsec * 1e-9;
}

static void env_setup(void)
{
	g_env_done = 1;
	const char *path = getenv("SOAPY_FAKE_FILE"), *e;
	if ((e = getenv("SOAPY_FAKE_PACE")) != NULL)
		g_pace = atof(e);
	if ((e = getenv("SOAPY_FAKE_MAX_READ")) != NULL && atol(e) > 0)
		g_max_chunk = (size_t)atol(e);
	if ((e = getenv("SOAPY_FAKE_EOF_SIGINT")) != NULL)
		g_eos_sigint_ms = atol(e);
	if (path && !g_src) {
		FILE *f = fopen(path, "rb");
		if (!f) {
			fprintf(stderr, "soapy_fake: cannot open %s\n", path);
			return;
		}
		fseek(f, 0, SEEK_END);
		long bytes = ftell(f);
		fseek(f, 0, SEEK_SET);
		int16_t *buf = malloc(bytes > 0 ? (size_t)bytes : 1);
		if (buf && fread(buf, 1, (size_t)bytes, f) == (size_t)bytes) {
			g_src = buf;
			g_src_len = (size_t)bytes / g_elem;
			g_src_pos = 0;
		}
		fclose(f);
	}
}

size_t SoapySDR_formatToSize(const char *format)
{
	/* bytes per element the way libSoapySDR counts them: all digits form the bit width of one component,
	 * doubled for complex formats, divided by 8 (CS16 -> 4, CS12 -> 3, CU8 -> 2, CF32 -> 8) */
	size_t bits = 0, is_complex = 0;
	for (const char *p = format; *p; p++) {
		if (*p == 'C') is_complex = 1;
		if (*p >= '0' && *p <= '9') bits = bits * 10 + (size_t)(*p - '0');
	}
	return (is_complex ? 2 : 1) * bits / 8;
}

SoapySDRKwargs SoapySDRKwargs_fromString(const char *markup) { SoapySDRKwargs k = {0, NULL, NULL}; (void)markup; return k; }
void SoapySDRKwargs_clear(SoapySDRKwargs *args) { if (args) args->size = 0; }

const char *SoapySDRDevice_lastError(void) { return "fake"; }
SoapySDRDevice *SoapySDRDevice_makeStrArgs(const char *args) { (void)args; return &g_dev; }
int SoapySDRDevice_unmake(SoapySDRDevice *d) { (void)d; return 0; }
char *SoapySDRDevice_getDriverKey(const SoapySDRDevice *d) { (void)d; return strdup("fake"); }
char *SoapySDRDevice_getHardwareKey(const SoapySDRDevice *d) { (void)d; return strdup("fake"); }
SoapySDRKwargs SoapySDRDevice_getHardwareInfo(const SoapySDRDevice *d) { SoapySDRKwargs k = {0, NULL, NULL}; (void)d; return k; }
size_t SoapySDRDevice_getNumChannels(const SoapySDRDevice *d, const int dir) { (void)d; (void)dir; return 1; }

SoapySDRStream *SoapySDRDevice_setupStream(SoapySDRDevice *d, const int dir, const char *fmt,
	const size_t *ch, const size_t n, const SoapySDRKwargs *a)
{ (void)d; (void)dir; (void)fmt; (void)ch; (void)n; (void)a; return &g_stream; }
int SoapySDRDevice_closeStream(SoapySDRDevice *d, SoapySDRStream *s) { (void)d; (void)s; return 0; }
int SoapySDRDevice_activateStream(SoapySDRDevice *d, SoapySDRStream *s, const int f, const long long t, const size_t n)
{ (void)d; (void)f; (void)t; (void)n; s->active = 1; return 0; }
int SoapySDRDevice_deactivateStream(SoapySDRDevice *d, SoapySDRStream *s, const int f, const long long t)
{ (void)d; (void)f; (void)t; s->active = 0; return 0; }

int SoapySDRDevice_readStream(SoapySDRDevice *d, SoapySDRStream *s, void * const *buffs,
	const size_t numElems, int *flags, long long *timeNs, const long timeoutUs)
{
	size_t n = numElems;
	(void)d; (void)s; (void)flags; (void)timeNs; (void)timeoutUs;
	if (!g_env_done)
		env_setup();
	if (g_discard && buffs[0] == g_discard) {
		memset(buffs[0], 0, n * 2 * sizeof(int16_t));
		return (int)n;
	}
	if (g_pace_hook && g_reads++)
		g_pace_hook();
	if (!g_src || g_src_pos >= g_src_len) {
		if (g_eos_hook) { void (*h)(void) = g_eos_hook; g_eos_hook = NULL; h(); }
		if (g_eos_sigint_ms >= 0) {
			long ms = g_eos_sigint_ms;
			g_eos_sigint_ms = -1;
			usleep((useconds_t)ms * 1000);
			raise(SIGINT);
		} else if (!g_pace_hook) {
			usleep(1000);                                  /* an exhausted capture is polled, not spun on */
		}
		return SOAPY_SDR_STREAM_ERROR;
	}
	if (g_max_chunk && n > g_max_chunk) n = g_max_chunk;
	if (n > g_src_len - g_src_pos) n = g_src_len - g_src_pos;
	if (g_pace > 0 && g_dev.rate > 0) {
		/* deliver no faster than pace x the sample rate the program asked for: a block is "received" one block time after the previous
		 * one was handed out -- a reader that stalled (the first full_demod of a process pages the HIP runtime in) does not get the
		 * backlog in a burst, which the reference's single-slot hand-off (rtl_fm.c:858-862) would lose */
		const double now = mono_s();
		if (g_t0 < now)
			g_t0 = now;
		g_t0 += (double)n / (g_dev.rate * g_pace);
		g_paced += n;
		if (g_t0 > now)
			usleep((useconds_t)((g_t0 - now) * 1e6));
	}
	memcpy(buffs[0], (const char *)g_src + g_elem * g_src_pos, n * g_elem);
	g_src_pos += n;
	return (int)n;
}

int SoapySDRDevice_setAntenna(SoapySDRDevice *d, const int dir, const size_t ch, const char *n) { (void)d; (void)dir; (void)ch; (void)n; return 0; }
static char **empty_list(size_t *length) { *length = 0; return NULL; }
char **SoapySDRDevice_listAntennas(const SoapySDRDevice *d, const int dir, const size_t ch, size_t *l) { (void)d; (void)dir; (void)ch; return empty_list(l); }
char **SoapySDRDevice_listGains(const SoapySDRDevice *d, const int dir, const size_t ch, size_t *l) { (void)d; (void)dir; (void)ch; return empty_list(l); }
int SoapySDRDevice_setGainMode(SoapySDRDevice *d, const int dir, const size_t ch, const bool a) { (void)d; (void)dir; (void)ch; (void)a; return 0; }
int SoapySDRDevice_setGain(SoapySDRDevice *d, const int dir, const size_t ch, const double v) { (void)d; (void)dir; (void)ch; (void)v; return 0; }
int SoapySDRDevice_setGainElement(SoapySDRDevice *d, const int dir, const size_t ch, const char *n, const double v) { (void)d; (void)dir; (void)ch; (void)n; (void)v; return 0; }
int SoapySDRDevice_setFrequency(SoapySDRDevice *d, const int dir, const size_t ch, const double f, const SoapySDRKwargs *a) { (void)dir; (void)ch; (void)a; d->freq = f; return 0; }
double SoapySDRDevice_getFrequency(const SoapySDRDevice *d, const int dir, const size_t ch)
{
	(void)dir; (void)ch;
	if (g_freq_script && g_freq_n)
		return g_freq_script[g_freq_pos++ % g_freq_n];
	return d->freq;
}
char **SoapySDRDevice_listFrequencies(const SoapySDRDevice *d, const int dir, const size_t ch, size_t *l) { (void)d; (void)dir; (void)ch; return empty_list(l); }
int SoapySDRDevice_setFrequencyCorrection(SoapySDRDevice *d, const int dir, const size_t ch, const double v) { (void)d; (void)dir; (void)ch; (void)v; return 0; }
int SoapySDRDevice_setSampleRate(SoapySDRDevice *d, const int dir, const size_t ch, const double r) { (void)dir; (void)ch; d->rate = r; return 0; }
double *SoapySDRDevice_listSampleRates(const SoapySDRDevice *d, const int dir, const size_t ch, size_t *l) { (void)d; (void)dir; (void)ch; *l = 0; return NULL; }
int SoapySDRDevice_setBandwidth(SoapySDRDevice *d, const int dir, const size_t ch, const double bw) { (void)dir; (void)ch; d->bw = bw; return 0; }
double SoapySDRDevice_getBandwidth(const SoapySDRDevice *d, const int dir, const size_t ch) { (void)dir; (void)ch; return d->bw; }
double *SoapySDRDevice_listBandwidths(const SoapySDRDevice *d, const int dir, const size_t ch, size_t *l) { (void)d; (void)dir; (void)ch; *l = 0; return NULL; }
int SoapySDRDevice_writeSetting(SoapySDRDevice *d, const char *k, const char *v) { (void)d; (void)k; (void)v; return 0; }
char *SoapySDRDevice_readSetting(const SoapySDRDevice *d, const char *k) { (void)d; (void)k; return strdup("true"); }
