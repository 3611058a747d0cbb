/* capi.cpp -- extern "C" surface of libsaugns_amd.so (include/saugns_amd.h). */
#include "../../include/saugns_amd.h"
#include "engine.h"
#include "hip_backend.h"
#include "capi_internal.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <chrono>
#include <exception>
#include <string>
#include <vector>

using sauengine::Backend;
using sauengine::Engine;

namespace {
thread_local std::string g_last_error;

void report(const char *where, const std::string &err) {
	g_last_error = err;
	/* same channel as sau_warning/sau_error in the reference (sau/error.c) */
	fprintf(stderr, "error [%s]: %s\n", where, err.c_str());
}
} /* namespace */

struct sauAmdBatch {
	Engine *engine;
	sauhip::HipBackend *hip; /* NULL when a test backend was injected */
};

/* The drop-in generator renders ahead of its caller: the reference host asks
 * for 11289 frames at a time (saugns.c:589-618), and a device round trip per
 * such call would cost more than the rendering. A host that keeps asking for what it asked for
 * before -- every host there is -- is handed the PCM of one larger
 * engine run piecewise (the run reproduces the reference's block lattice for calls of that size:
 * engine.h, set_call_len), from three page-locked buffers: while the
 * host consumes one, the device renders and copies the next two runs into the
 * others -- two, so that a run is already queued when the one before it ends
 * (with one run in flight the device stood idle between a run's completion and
 * the host's next issue: 1.27 ms per 176400-frame run of BASELINE config 3
 * against 0.93 ms of kernels; SAU_AMD_READAHEAD_DEPTH=1 gives that back).
 * SAU_AMD_READAHEAD=<frames> sets the size of a run, 0 turns the scheme off.
 * sauGenerator_run takes `buf_len` and `stereo` per call, though (sau/generator.c:905-913), and frames rendered ahead for
 * calls of one size and layout are not what the reference gives a host that then asks for another: the lattice of <= 1024-
 * frame blocks restarts at every call (generator.c:854-878), and mono is (L + R) / 2 before rounding. So every run starts
 * from a snapshot of the engine (Engine::snapshot: host mirrors + the operator records on the device); a call that differs
 * from the ones the buffered runs were issued for takes the engine back to the start of the run being handed out, renders
 * what the host has consumed of it once more (discarded) and goes on from there in the new size / layout (round 5). */
struct sauGenerator {
	sauAmdBatch batch;
	static constexpr int SLOTS = 3;
	int16_t *slot[SLOTS] = {nullptr, nullptr, nullptr}; /* backend->alloc_host() */
	size_t slot_cap[SLOTS] = {0, 0, 0};    /* int16 values */
	size_t ahead_frames = 176400;          /* frames per engine run */
	int cur = 0;                           /* the slot being handed out */
	size_t pos = 0, len = 0;               /* its unread part, in frames */
	int queued = 0;                        /* runs issued into slot[cur + 1 ..] and not yet handed out (0..depth) */
	int depth = 2;                         /* runs in flight at most (SAU_AMD_READAHEAD_DEPTH, 1 or 2) */
	size_t q_len[SLOTS] = {0, 0, 0};       /* frames of the run in each slot */
	bool ahead_stereo = false;
	bool more = true;                      /* the engine has signal left after everything issued */
	/* The first runs are short and grow fourfold from one to the next (1, 4, 16, ... host calls) until they reach
	 * ahead_frames: what the host waits for in its first call is the events at t = 0 and one call's worth of
	 * rendering, not a whole read-ahead run (BASELINE config 5: a 176400-frame run is 25 ms of feedback chains).
	 * The device is never idle meanwhile -- the next run is always issued before the current one is handed out. */
	unsigned runs_issued = 0;
	bool ramp = true;                      /* SAU_AMD_READAHEAD_RAMP=0: every run ahead_frames long */
	unsigned grow_bits = 2;                /* a run is 2^grow_bits times the one before (SAU_AMD_READAHEAD_GROW) */
	/* Round 6: how long the runs are follows what the script costs. Every run has a fixed cost (about 0.15 ms of launches and
	 * bookkeeping on the device: BASELINE config 3's 10 s took 3.34 ms in runs of 1, 4, 15, 15, 5 calls and 2.68 ms as one run,
	 * profiles/r06_ab.txt), so a script is rendered in as few runs as the host's first wait allows: the first run is what an
	 * estimate of the script's cost per frame puts at FIRST_RUN_NS = 8 ms (whole calls, at least one), the runs after it grow
	 * 2^grow_bits-fold up to what it puts at LATER_RUN_NS (at least ahead_frames, at most MAX_RUN_FRAMES and what 4 GiB of
	 * voice rows hold). The estimate is deliberately pessimistic (closed-form voices run four times faster): a script too
	 * heavy for it starts, as before, with one call. SAU_AMD_READAHEAD=<frames> fixes the later runs' length instead. */
	double est_ns_frame = 0;               /* estimated device time per frame of this script (make_generator) */
	size_t run_cap = 0;                    /* frames per run at most (memory), 0: none */
	bool ahead_set = false;                /* SAU_AMD_READAHEAD given: the later runs are ahead_frames long */
	size_t ahead_call = 0;                 /* the call size the buffered runs were issued for (their block lattice) */
	Engine::Snapshot snap[SLOTS];          /* the engine before the run in each slot */
	unsigned rewinds = 0;                  /* times a changed call took the engine back (sauAmd_Generator_rewinds: tests) */
};

/* sau/generator/noise.h:18-21 */
extern "C" const char *const sauNoise_names[SAU_NOISE_NAMED + 1] = {
	"wh", "gw", "bw", "tw", "re", "vi", "bv", nullptr
};

static bool make_batch(sauAmdBatch &b, const sauProgram *const *prgs, size_t n,
		uint32_t srate, Backend *injected, int device = -1) {
	std::string err;
	b.engine = nullptr;
	b.hip = nullptr;
	Backend *be = injected;
	const auto t0 = std::chrono::steady_clock::now();
	if (!be) {
		b.hip = sauhip::create_hip_backend(err, device);
		if (!b.hip) { report("generator", err); return false; }
		be = b.hip;
	}
	const auto t1 = std::chrono::steady_clock::now();
	for (size_t i = 0; i < n; ++i)
		if (!prgs[i]) { report("generator", "NULL program"); delete be; b.hip = nullptr; return false; }
	b.engine = Engine::create(prgs, n, srate, be, err);
	if (!b.engine) { b.hip = nullptr; report("generator", err); return false; }
	if (getenv("SAU_AMD_DEBUG_CREATE")) {
		const auto t2 = std::chrono::steady_clock::now();
		fprintf(stderr, "saugns_amd: backend object %.3f ms, engine (program conversion, device state) %.3f ms\n",
				std::chrono::duration<double, std::milli>(t1 - t0).count(),
				std::chrono::duration<double, std::milli>(t2 - t1).count());
	}
	return true;
}

sauGenerator *sauamd_internal::make_generator(const sauProgram *prg, uint32_t srate, Backend *injected) {
	if (!prg) return nullptr;
	sauGenerator *g = new sauGenerator();
	if (!make_batch(g->batch, &prg, 1, srate, injected)) { delete g; return nullptr; }
	if (const char *ra = getenv("SAU_AMD_READAHEAD")) { g->ahead_frames = (size_t)atol(ra); g->ahead_set = true; }
	{ /* the script's cost per frame, from what its program says: every operator as if it ran all the time, at the rate of
	   * the slower time-parallel builds; a self-modulated oscillator anywhere makes every frame a step of a recurrence */
		bool feedback = false;
		for (size_t e = 0; e < prg->ev_count && !feedback; ++e)
			for (uint32_t d = 0; d < prg->events[e].op_data_count; ++d) {
				const sauLine *pm = prg->events[e].op_data[d].pm_a;
				if (pm && (((pm->flags & SAU_LINEP_STATE) && pm->v0 != 0.f) || ((pm->flags & SAU_LINEP_GOAL) && pm->vt != 0.f))) { feedback = true; break; }
			}
		g->est_ns_frame = 1.0 + 0.004 * (double)prg->op_count + (feedback ? 120.0 : 0.0);
		const size_t vo = prg->vo_count ? prg->vo_count : 1;
		g->run_cap = ((size_t)1 << 30) / vo; /* (4 GiB of f32 voice rows) */
		const size_t script = (size_t)((uint64_t)prg->duration_ms * srate / 1000) + 1; /* (no run needs to be longer than the script) */
		if (script < g->run_cap) g->run_cap = script;
	}
	if (const char *rr = getenv("SAU_AMD_READAHEAD_RAMP")) g->ramp = atoi(rr) != 0;
	if (const char *rg = getenv("SAU_AMD_READAHEAD_GROW")) { const int b = atoi(rg); if (b >= 1 && b <= 8) g->grow_bits = (unsigned)b; }
	if (const char *rd = getenv("SAU_AMD_READAHEAD_DEPTH")) g->depth = atoi(rd) >= 2 ? 2 : 1;
	return g;
}

extern "C" sauGenerator *sau_create_Generator(const sauProgram *prg, uint32_t srate) {
	return sauamd_internal::make_generator(prg, srate, nullptr);
}
unsigned sauamd_internal::generator_rewinds(const sauGenerator *g) { return g ? g->rewinds : 0u; }

extern "C" void sau_destroy_Generator(sauGenerator *o) {
	if (!o) return;
	Backend *be = o->batch.engine->backend();
	std::string err;
	(void)be->sync(err); /* a queued run may still be writing into a slot */
	for (int i = 0; i < sauGenerator::SLOTS; ++i) be->free_host(o->slot[i]);
	delete o->batch.engine;
	delete o;
}

static bool generator_fail(sauGenerator *o, int16_t *buf, size_t buf_len, bool stereo,
		size_t *out_len, const std::string &err) {
	/* the reference cannot fail here: report, give silence, end */
	report("generator", err);
	memset(buf, 0, sizeof(int16_t) * buf_len * (stereo ? 2 : 1));
	if (out_len) *out_len = 0;
	o->more = false; o->pos = o->len = 0; o->queued = 0;
	return false;
}

/* Start the next engine run; its PCM lands in the first slot behind the one being handed out and those queued. */
static bool generator_issue(sauGenerator *o, size_t big, size_t call_len, bool stereo, std::string &err) {
	const int k = (o->cur + 1 + o->queued) % sauGenerator::SLOTS;
	const size_t ch = stereo ? 2 : 1;
	Backend *be = o->batch.engine->backend();
	/* this run: whole host calls, four times as many as the run before, up to `big` (every run has a fixed cost of
	 * about 0.1 ms: doubling cost the 95 corpus scripts 40 ms of their 410) */
	size_t frames = big;
	if (o->ramp && call_len) {
		/* the first run: what the estimate puts at FIRST_RUN_NS, in whole calls (at least one); then 2^grow_bits-fold from run to run */
		constexpr double FIRST_RUN_NS = 8e6; /* (estimated; a closed-form bank renders four times faster: BASELINE config 3's whole 10 s are one run of 2.5 ms) */
		size_t first = (size_t)(FIRST_RUN_NS / o->est_ns_frame) / call_len * call_len;
		if (first < call_len) first = call_len;
		const unsigned sh = o->grow_bits * o->runs_issued;
		if (sh < 24 && first < (big >> sh)) frames = first << sh;
	}
	++o->runs_issued;
	if (o->runs_issued == 1 && !o->batch.engine->reserve(big, stereo, err)) return false; /* (device buffers too) */
	if (o->slot_cap[k] < big * ch) { /* (sized for the longest run at once: growing later would wait for the stream) */
		be->free_host(o->slot[k]);
		o->slot_cap[k] = 0;
		o->slot[k] = (int16_t *)be->alloc_host(big * ch * sizeof(int16_t));
		if (!o->slot[k]) { err = "out of page-locked memory"; return false; }
		o->slot_cap[k] = big * ch;
	}
	bool more = false;
	size_t len = 0;
	if (!sauengine::tune_env("SAU_AMD_NO_SNAPSHOT") && /* (debugging aid: a later change of the calls then fails) */
	    !o->batch.engine->snapshot(o->snap[k], k, err)) return false; /* (where a call of another size or layout goes back to) */
	/* PCM stays on the device; the copy queues behind the mixer */
	if (!o->batch.engine->run(nullptr, frames, stereo, &more, &len, err)) return false;
	if (len && !be->fetch_pcm_async(0, o->slot[k], (uint32_t)len, stereo, k, err)) return false;
	++o->queued;
	o->q_len[k] = len;
	o->more = more;
	o->ahead_stereo = stereo;
	o->ahead_call = call_len;
	return true;
}

/* The host's call is not of the kind the buffered runs were rendered for: back to the start of the run being handed out
 * (or, when that one is used up, of the first one queued), what the host has had of it rendered once more in the old size
 * and layout -- the operators then stand where the reference's do after those calls -- and everything queued dropped. */
static bool generator_rewind(sauGenerator *o, std::string &err) {
	Backend *be = o->batch.engine->backend();
	Engine *en = o->batch.engine;
	int k = o->cur;
	size_t consumed = o->pos;
	if (o->pos == o->len) { /* (then something is queued: the caller has checked) */
		k = (o->cur + 1) % sauGenerator::SLOTS;
		consumed = 0;
	}
	if (!be->sync(err)) return false; /* the runs in flight end where they end; their PCM is dropped */
	if (!en->restore(o->snap[k], k, err)) return false;
	o->more = true; /* (a run is only ever issued while there is signal left) */
	if (consumed) {
		bool more = false;
		size_t len = 0;
		en->set_call_len(o->ahead_call);
		if (!en->run(nullptr, consumed, o->ahead_stereo, &more, &len, err)) return false;
		o->more = more;
	}
	o->pos = o->len = 0;
	o->queued = 0;
	o->runs_issued = 0; /* short runs first again: the host is waiting */
	++o->rewinds;
	return true;
}

static bool generator_run(sauGenerator *o, int16_t *buf, size_t buf_len, bool stereo, size_t *out_len);
/* No C++ exception may cross the C ABI (the hosts are C): an allocation that fails in mid-run ends the render the way a
 * backend error does -- a message, silence, false. */
extern "C" bool sauGenerator_run(sauGenerator *o, int16_t *buf, size_t buf_len,
		bool stereo, size_t *out_len) {
	try {
		return generator_run(o, buf, buf_len, stereo, out_len);
	} catch (const std::exception &ex) {
		return generator_fail(o, buf, buf_len, stereo, out_len, std::string("internal error: ") + ex.what());
	}
}
static bool generator_run(sauGenerator *o, int16_t *buf, size_t buf_len, bool stereo, size_t *out_len) {
	std::string err;
	const size_t ch = stereo ? 2 : 1;
	if (buf_len == 0) {
		if (out_len) *out_len = 0;
		return o->more || o->queued || o->pos < o->len;
	}
	if (o->pos == o->len && !o->queued && (o->ahead_frames == 0 || buf_len >= o->ahead_frames)) {
		/* nothing buffered and the call is large: render straight into the caller's buffer */
		bool more = false;
		size_t len = 0;
		int16_t *bufs[1] = {buf};
		if (!o->more) { memset(buf, 0, sizeof(int16_t) * buf_len * ch); if (out_len) *out_len = 0; return false; }
		o->batch.engine->set_call_len(0); /* this run is the host's call */
		if (!o->batch.engine->run(bufs, buf_len, stereo, &more, &len, err))
			return generator_fail(o, buf, buf_len, stereo, out_len, err);
		o->more = more;
		if (out_len) *out_len = len;
		return more;
	}
	/* An engine run covers whole host calls, so that it starts where one of them does: the
	 * reference's block lattice restarts at every call (generator.c:854-878; Lattice in
	 * sau_dev_math.h), and the engine lays it out for calls of this size. A host that changes its
	 * call size or its channel layout mid-stream (saugns.c never does; the interface allows both) gets what the
	 * reference would give it: the frames buffered for the old kind of call are rendered again. */
	if ((o->pos < o->len || o->queued) && (o->ahead_stereo != stereo || o->ahead_call != buf_len) &&
	    !generator_rewind(o, err))
		return generator_fail(o, buf, buf_len, stereo, out_len, err);
	if (o->pos == o->len && !o->queued && (o->ahead_frames == 0 || buf_len >= o->ahead_frames))
		return generator_run(o, buf, buf_len, stereo, out_len); /* (nothing buffered any more and the call is large: straight into the caller's buffer, above) */
	size_t big = buf_len >= o->ahead_frames ? buf_len : o->ahead_frames / buf_len * buf_len;
	if (!o->ahead_set && o->ramp && o->est_ns_frame > 0) { /* (see sauGenerator::est_ns_frame) */
		constexpr double LATER_RUN_NS = 20e6;
		constexpr size_t MAX_RUN_FRAMES = (size_t)1 << 20;
		size_t want = (size_t)(LATER_RUN_NS / o->est_ns_frame);
		if (want > MAX_RUN_FRAMES) want = MAX_RUN_FRAMES;
		if (o->run_cap && want > o->run_cap) want = o->run_cap;
		want = (want + buf_len - 1) / buf_len * buf_len;
		if (want > big) big = want;
	}
	o->batch.engine->set_call_len(buf_len);
	size_t filled = 0;
	bool issued = false;
	while (filled < buf_len) {
		if (o->pos == o->len) {
			if (!o->queued) {
				if (!o->more) break;
				if (!generator_issue(o, big, buf_len, stereo, err))
					return generator_fail(o, buf, buf_len, stereo, out_len, err);
			}
			const int nx = (o->cur + 1) % sauGenerator::SLOTS;
			if (!o->batch.engine->backend()->wait_fetch(nx, err))
				return generator_fail(o, buf, buf_len, stereo, out_len, err);
			o->cur = nx;
			o->pos = 0; o->len = o->q_len[nx]; --o->queued;
			/* the device goes on with the run after this one while the host consumes */
			if (o->more && o->queued < o->depth) {
				if (!generator_issue(o, big, buf_len, stereo, err))
					return generator_fail(o, buf, buf_len, stereo, out_len, err);
				issued = true;
			}
			if (o->len == 0) { if (!o->queued) break; continue; }
		}
		size_t n = o->len - o->pos;
		if (n > buf_len - filled) n = buf_len - filled;
		memcpy(buf + filled * ch, o->slot[o->cur] + o->pos * ch, n * ch * sizeof(int16_t));
		filled += n;
		o->pos += n;
	}
	if (filled < buf_len) /* generator.c:911-914: the rest of the buffer is silence */
		memset(buf + filled * ch, 0, (buf_len - filled) * ch * sizeof(int16_t));
	/* a second run in flight, so that one is queued when the one before it ends -- topped up by a call that issued
	 * nothing itself (never the first: what the host waits for there is one call's worth of rendering) */
	if (!issued && o->more && o->queued >= 1 && o->queued < o->depth && o->ahead_stereo == stereo)
		if (!generator_issue(o, big, buf_len, stereo, err))
			return generator_fail(o, buf, buf_len, stereo, out_len, err);
	if (out_len) *out_len = filled;
	return o->more || o->queued || o->pos < o->len;
}

extern "C" sauAmdBatch *sauAmd_create_Batch(const sauProgram *const *prgs, size_t n,
		uint32_t srate) {
	return sauamd_internal::make_batch_over(prgs, n, srate, nullptr);
}

sauAmdBatch *sauamd_internal::make_batch_over(const sauProgram *const *prgs, size_t n, uint32_t srate, Backend *injected, int device) {
	if (!prgs || !n) return nullptr;
	sauAmdBatch *b = new sauAmdBatch();
	if (!make_batch(*b, prgs, n, srate, injected, device)) { delete b; return nullptr; }
	return b;
}

/* include/saugns_amd.h: a batch on HIP device `device` of this process (a C host that spreads independent renders over the
 * node's GPUs: one batch per device, each with its own stream, buffers and chain-row budget; no device < 0) */
extern "C" sauAmdBatch *sauAmd_create_Batch_on(int device, const sauProgram *const *prgs, size_t n, uint32_t srate) {
	if (device < 0) { report("generator", "sauAmd_create_Batch_on: no device " + std::to_string(device)); return nullptr; }
	return sauamd_internal::make_batch_over(prgs, n, srate, nullptr, device);
}

extern "C" void sauAmd_destroy_Batch(sauAmdBatch *b) {
	if (!b) return;
	delete b->engine;
	delete b;
}

extern "C" bool sauAmd_Batch_run(sauAmdBatch *b, int16_t *const *bufs, size_t buf_len,
		bool stereo, bool *more, size_t *out_len) {
	std::string err;
	try {
		if (!b->engine->run(bufs, buf_len, stereo, more, out_len, err)) {
			report("batch", err);
			return false;
		}
	} catch (const std::exception &ex) { /* (nothing C++ crosses the C ABI) */
		report("batch", std::string("internal error: ") + ex.what());
		return false;
	}
	return true;
}

extern "C" void sauAmd_Batch_set_call_len(sauAmdBatch *b, size_t frames) {
	b->engine->set_call_len(frames);
}

extern "C" const int16_t *sauAmd_Batch_device_pcm(sauAmdBatch *b, size_t stream) {
	return b->engine->backend()->device_pcm((uint32_t)stream);
}

extern "C" bool sauAmd_Batch_sync(sauAmdBatch *b) {
	std::string err;
	if (!b->engine->backend()->sync(err)) { report("batch", err); return false; }
	return true;
}

extern "C" void sauAmd_Batch_timing(sauAmdBatch *b, double *render_ms, double *mix_ms,
		uint64_t *render_launches, int reset) {
	if (render_ms) *render_ms = 0;
	if (mix_ms) *mix_ms = 0;
	if (render_launches) *render_launches = 0;
	if (b->hip) b->hip->timing(render_ms, mix_ms, render_launches, reset != 0);
}

extern "C" void sauAmd_Batch_timing_ex(sauAmdBatch *b, double *out4, uint64_t *segments, int reset) {
	out4[0] = out4[1] = out4[2] = out4[3] = 0;
	if (segments) *segments = 0;
	if (b->hip) b->hip->timing_ex(out4, segments, reset != 0);
}

extern "C" void sauAmd_Batch_set_timing(sauAmdBatch *b, int level) {
	if (b->hip) b->hip->set_timing(level);
}

extern "C" bool sauAmd_Batch_order_after(sauAmdBatch *b, sauAmdBatch *before) {
	if (!b || !before || !b->hip || !before->hip) return true; /* (injected backends render synchronously: nothing to order) */
	std::string err;
	if (!b->hip->order_after(before->hip, err)) { report("batch", err); return false; }
	return true;
}

extern "C" void *sauAmd_Batch_stream(sauAmdBatch *b) {
	return b->hip ? b->hip->stream_handle() : nullptr;
}

extern "C" void sauAmd_set_piluts(const float *tables) {
	if (tables) sauengine::override_piluts(tables);
}
extern "C" const float *sauAmd_get_piluts(void) { return sauengine::builtin_piluts(); }

extern "C" const char *sauAmd_last_error(void) { return g_last_error.c_str(); }

extern "C" int sauAmd_device_count(void) { return sauhip::device_count(); }
extern "C" bool sauAmd_device_pci_bus_id(int device, char *buf, size_t len) {
	return sauhip::device_pci_bus_id(device, buf, (int)(len > 255 ? 255 : len));
}
