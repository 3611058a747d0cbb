/* saugns_amd.h -- C ABI of the MI355X generator backend (libsaugns_amd.so).
 *
 * Part 1 is the drop-in boundary: exactly the symbols sau/generator.o exports
 * in the reference, with the same argument meaning and
 * error behaviour.  Linking this library ahead of -lsau makes the unchanged
 * reference host (parser, player) render on the GPU; see INTEGRATION.md.
 *
 * Part 2 are extensions around the same hot path: many programs rendered in
 * lock step (BASELINE config 4), device-resident PCM, wave-table injection,
 * and a pointer-free program image so that programs can be stored and
 * rebuilt without the reference parser.
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 */
#ifndef SAUGNS_AMD_H
#define SAUGNS_AMD_H

#include "sau_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SAU_AMD_API __attribute__((visibility("default")))

/* ---- Part 1: drop-in boundary ------------------------------------------- */

typedef struct sauGenerator sauGenerator;

/* replaces sau/generator.h:20-21 (generator.c:200-217).
 * NULL on failure (no usable GPU, out of memory, graph too large). */
SAU_AMD_API sauGenerator *sau_create_Generator(const sauProgram *prg, uint32_t srate);

/* replaces sau/generator.h:22 (generator.c:222-228). NULL-safe. */
SAU_AMD_API void sau_destroy_Generator(sauGenerator *o);

/* replaces sau/generator.h:24-26 (generator.c:905-973): fill buf with
 * buf_len frames (interleaved L,R when stereo, else (L+R)/2); returns false
 * once the signal has ended, with *out_len = frames generated this call. */
SAU_AMD_API bool sauGenerator_run(sauGenerator *o, int16_t *buf, size_t buf_len,
		bool stereo, size_t *out_len);

/* replaces the definition in sau/generator/noise.h:18-21, which the
 * reference's parser.o and help.o import. */
SAU_AMD_API extern const char *const sauNoise_names[SAU_NOISE_NAMED + 1];

/* ---- Part 2: extensions --------------------------------------------------- */

typedef struct sauAmdBatch sauAmdBatch;

/* n independent programs rendered together; stream i renders prgs[i].
 * Programs are borrowed and must outlive the batch. NULL on failure. */
SAU_AMD_API sauAmdBatch *sauAmd_create_Batch(const sauProgram *const *prgs, size_t n,
		uint32_t srate);
/* The same on HIP device `device` (0 <= device < sauAmd_device_count()) whatever SAU_AMD_DEVICE says: a host that shards
 * independent renders over the GPUs of a node from one process (contiguous blocks of renders per GPU, no exchange
 * step) creates one batch per device and runs them side by side -- every batch has its own stream, buffers, pools and budget
 * for the feedback chains' rows. NULL on failure (sauAmd_last_error), also for a device that does not exist. */
SAU_AMD_API sauAmdBatch *sauAmd_create_Batch_on(int device, const sauProgram *const *prgs, size_t n,
		uint32_t srate);
SAU_AMD_API void sauAmd_destroy_Batch(sauAmdBatch *b);

/* Advance every stream by buf_len frames. bufs may be NULL, or hold one host
 * pointer per stream (entries may be NULL): PCM is copied there; otherwise it
 * stays on the device. more[i]/out_len[i] (either may be NULL) as
 * sauGenerator_run for stream i. Returns false on a backend error. */
SAU_AMD_API bool sauAmd_Batch_run(sauAmdBatch *b, int16_t *const *bufs, size_t buf_len,
		bool stereo, bool *more, size_t *out_len);

/* The reference restarts its <= 1024-frame rendering blocks at every sauGenerator_run call
 * (generator.c:854-878), and the positions of held lines move block by block (sau/line.c:385-398),
 * so a render depends -- in rare event sequences -- on the caller's buffer size. By default each
 * sauAmd_Batch_run call stands for one such call. frames > 0: a run covers consecutive calls of
 * that many frames each (rendering far ahead of a host that asks for 11289 frames at a time). */
SAU_AMD_API void sauAmd_Batch_set_call_len(sauAmdBatch *b, size_t frames);

/* Device address of stream i's PCM row of the last run (hipMalloc memory). */
SAU_AMD_API const int16_t *sauAmd_Batch_device_pcm(sauAmdBatch *b, size_t stream);

/* Wait for all queued device work of the batch. */
SAU_AMD_API bool sauAmd_Batch_sync(sauAmdBatch *b);

/* Accumulated HIP-event timings of the render / mix kernels since the last
 * call with reset != 0 (any pointer may be NULL). Enables timing on first use. */
SAU_AMD_API void sauAmd_Batch_timing(sauAmdBatch *b, double *render_ms, double *mix_ms,
		uint64_t *render_launches, int reset);

/* Per-kernel split of the same timings: out4[0] time-parallel kernel
 * (fast_kernel), [1] block-loop kernel (render_kernel), [2] mixer, [3] analyze
 * + finalize, all in ms; *segments = number of rendered segments. */
SAU_AMD_API void sauAmd_Batch_timing_ex(sauAmdBatch *b, double *out4, uint64_t *segments,
		int reset);

/* HIP-event timing: 0 off, 1 only the dominant (time-parallel) kernel,
 * 2 every kernel. The query functions above switch level 2 on when still off. */
SAU_AMD_API void sauAmd_Batch_set_timing(sauAmdBatch *b, int level);

/* The stream the batch launches its kernels on, as a hipStream_t. */
SAU_AMD_API void *sauAmd_Batch_stream(sauAmdBatch *b);

/* Order b's next run behind what has been issued for `before` so far: the rendering kernels of b's next
 * sauAmd_Batch_run start on the device only when everything queued for `before` has finished (b's bookkeeping --
 * operator updates, plan uploads, the per-voice analysis -- may run earlier: it touches nothing of `before`'s).
 * A host that renders one script after another with two generators alive (saugns.c:583-621 per script) issues
 * script k + 1 while script k's last kernels drain: the host work of a run's start (events at t = 0, plans: 1.2 ms
 * for BASELINE config 5's 4096 voices) overlaps the device's tail, and the two scripts' kernels never compete for
 * the device -- two runs simply issued side by side did, at three times the time per script. Both on the same
 * device. false (sauAmd_last_error) on a HIP error. */
SAU_AMD_API bool sauAmd_Batch_order_after(sauAmdBatch *b, sauAmdBatch *before);

/* Use these twelve 2048-entry tables (wave-id order) instead of the built-in
 * ones for generators created afterwards. */
SAU_AMD_API void sauAmd_set_piluts(const float *tables);
/* The tables a new generator would use. */
SAU_AMD_API const float *sauAmd_get_piluts(void);

/* Text of the last error on this thread ("" if none). */
SAU_AMD_API const char *sauAmd_last_error(void);

/* Number of visible HIP devices; <= 0 when none or HIP is unusable. */
SAU_AMD_API int sauAmd_device_count(void);
/* PCI address ("0000:c1:00.0") of HIP device `device` into buf (len >= 16); false when there is no such device. One
 * process per GPU (SAU_AMD_DEVICE selects it): a multi-rank job can show that its ranks sit on distinct boards. */
SAU_AMD_API bool sauAmd_device_pci_bus_id(int device, char *buf, size_t len);

/* Program image: every struct of a sauProgram in one relocatable block.
 * serialize returns the size needed; nothing is written if cap is smaller. */
SAU_AMD_API size_t sauAmd_program_serialize(const sauProgram *prg, void *buf, size_t cap);
/* Rebuild a sauProgram from an image (one allocation; free with _free). */
SAU_AMD_API sauProgram *sauAmd_program_load(const void *image, size_t len);
SAU_AMD_API void sauAmd_program_free(sauProgram *prg);

/* Voice banks without the parser (for a 1024-voice bank sau_build_Program,
 * sau/parser.c:2092, takes longer than the whole render here). Operators are given as a flat array: carriers
 * (use == SAU_POP_N_carr, one voice each, in time order) and modulators naming their parent's index and how it
 * uses them. The result is laid out as the parser lays out the equivalent script -- one event per voice,
 * operator data in post-order, ids in pre-order, line and time flags of freshly created operators -- and is
 * accepted by sau_create_Generator / sauAmd_create_Batch like any other program. */
typedef struct sauAmdLineDesc {
	uint8_t present;   /* 0: the parameter is not given (the generator's default applies) */
	uint8_t has_goal;  /* a sweep to `goal` over the operator's time */
	uint8_t ratio;     /* frequency lines: value is a ratio of the parent's frequency */
	uint8_t shape;     /* SAU_LINE_N_* */
	float v0, goal;
} sauAmdLineDesc;
typedef struct sauAmdOpDesc {
	uint32_t parent;   /* index of the operator this one modulates (ignored for carriers) */
	uint32_t use;      /* SAU_POP_N_* */
	uint32_t type;     /* SAU_POPT_N_* */
	uint32_t mode;     /* W: wave id; N: noise id; R: line | function flags << 8 | function << 16 */
	uint32_t time_ms;  /* 0: implicit (lasts as long as its parent), lines then take default_mod_ms */
	uint32_t start_ms; /* carriers: when the voice begins */
	uint32_t phase;    /* cycle fraction, 2^32 = one turn */
	uint32_t seed;
	sauAmdLineDesc pan, amp, amp2, freq, freq2, pm_a;
} sauAmdOpDesc;
/* NULL on a malformed description: parent out of range, cycles, voices out of time order, an operator type,
 * wave / noise id, R line or function or a ramp shape outside its enum, a carrier with time_ms == 0 (a voice's
 * length is its carrier's), nesting deeper than 255, a duration beyond 32 bits of milliseconds. */
SAU_AMD_API sauProgram *sauAmd_build_bank(const sauAmdOpDesc *ops, size_t n_ops, float ampmult,
		uint32_t default_mod_ms);
SAU_AMD_API void sauAmd_free_bank(sauProgram *prg);

/* Output stage (replaces the reference's player/sndfile.c:125-210 writer fed
 * from Player_run's chunk loop, saugns.c:589-618): render prg to a sound file.
 * format as SGS_SNDFILE_* (player/sndfile.h:21-26); channels 1 or 2. The file
 * is byte-identical to what the reference writer produces from the same PCM.
 * *frames_out (may be NULL) = frames written. False on failure. */
enum { SAU_AMD_SNDFILE_RAW = 0, SAU_AMD_SNDFILE_AU = 1, SAU_AMD_SNDFILE_WAV = 2 };
SAU_AMD_API bool sauAmd_render_file(const sauProgram *prg, uint32_t srate, const char *path,
		int format, int channels, uint64_t *frames_out);

#ifdef __cplusplus
}
#endif
#endif /* SAUGNS_AMD_H */
