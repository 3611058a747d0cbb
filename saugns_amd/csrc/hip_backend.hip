/* hip_backend.hip -- gfx950 kernels and the HIP backend of the generator.
 *
 * One translation unit (the kernels share parameter blocks and inlined helpers), in parts:
 *   k_common.h      parameter blocks of the block loop, wave-level helpers (DPP scans, look-back words)
 *   k_block_loop.h  render_kernel<W,T,V>
 *   k_fast_types.h  FastParams, FastInfo, constants of the time-parallel path
 *   k_analyze.h     analyze_kernel
 *   k_decode.h      FastStep / FastLine / FastAux / ChainDesc, scan_kernel, decode_kernel
 *   k_fast_voice.h  fast_voice, fast_kernel<T,SCAN>, repair_kernel<T>
 *   k_chain.h       chain_kernel
 *   k_finish.h      finalize_kernel, mix_kernel, event_kernel
 *   (this file)     buffer and stream pools, table sets, HipBackendImpl: what is launched when
 *
 * Work decomposition (DESIGN.md "Kernels"):
 *   render_kernel<W,T>  one workgroup (W waves) per live voice.  The voice's
 *       operator states (256 B each) and all block buffers live in LDS for
 *       the whole segment; Hermite coefficient tables of the wave types in
 *       use are staged into LDS once per workgroup.  Time runs in blocks of
 *       W*(64T-1) samples; inside a block every lane owns T consecutive
 *       samples, so the differentiator's "previous sample" is in-lane except
 *       at lane boundaries (one cross-lane shuffle) and wave boundaries (one
 *       overlapping halo sample per wave instead of an exchange).  Phase
 *       accumulation is an exact integer prefix scan (wave shuffles + one LDS
 *       exchange).  The only serial code is the feedback recurrence of
 *       self-modulating operators and the rare dphase==0 fill-forward.
 *       Each voice's carrier block goes to HBM once (f32 [voice][frame]).
 *   mix_kernel  one thread per output frame sums the voices of its stream in
 *       ascending voice order (the reference's f32 accumulation order,
 *       generator.c:749-788) and writes int16 PCM (795-825).
 *   event_kernel  applies operator updates to the state in HBM.
 *
 * Arithmetic: sau_dev_math.h, compiled with -ffp-contract=off.
 */
#include <hip/hip_runtime.h>
#include "hip_backend.h"
#include "sau_dev_ops.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

namespace sauhip {

using namespace saudev;
using sauengine::BackendConfig;
using sauengine::tune_env;
using sauengine::SegmentDesc;

/* ------------------------------------------------------------------------ */
/* device side                                                              */
/* ------------------------------------------------------------------------ */
#include "k_common.h"
#include "k_block_loop.h"
#include "k_fast_types.h"
#include "k_analyze.h"
#include "k_decode.h"
#include "k_fast_voice.h"
#include "k_chain.h"
#include "k_finish.h"

/* ------------------------------------------------------------------------ */
/* host side                                                                */
/* ------------------------------------------------------------------------ */

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
	err = std::string(#call) + ": " + hipGetErrorString(e_); return false; } } while (0)

/* Device and page-locked buffers are recycled through a process-wide pool: a host that renders one
 * script after another (saugns.c:583-621 once per script) would otherwise pay about twenty
 * hipMalloc/hipFree pairs, each a device-wide synchronisation, per generator. Blocks are handed
 * back only after the owning stream has drained (the destructor and grow() see to that). */
class BufPool {
public:
	static BufPool &get() { static BufPool *g = new BufPool; return *g; } /* never destroyed: HIP may be gone by then */
	static size_t bucket(size_t bytes) {
		size_t b = 4096;
		while (b * 2 <= bytes) b <<= 1;
		const size_t q = b / 8; /* eight size classes per octave */
		return (bytes + q - 1) / q * q;
	}
	void *take(bool pinned, size_t bytes) {
		int dev = 0;
		(void)hipGetDevice(&dev);
		std::lock_guard<std::mutex> lk(mu_);
		auto &m = pinned ? pin_ : dev_[dev & 15];
		auto it = m.find(bytes);
		if (it == m.end()) return nullptr;
		void *q = it->second;
		m.erase(it);
		(pinned ? held_pin_ : held_dev_) -= bytes;
		return q;
	}
	/* false: the pool is full, the caller frees the block */
	bool give(bool pinned, void *q, size_t bytes) {
		int dev = 0;
		(void)hipGetDevice(&dev);
		std::lock_guard<std::mutex> lk(mu_);
		size_t &held = pinned ? held_pin_ : held_dev_;
		/* what the pool may keep between generators: 64 GiB of the 288 GB of HBM, 1 GiB page-locked;
		 * SAU_AMD_POOL_MB / SAU_AMD_PINNED_POOL_MB set other caps (0: keep nothing) */
		static const size_t cap_dev = env_mb("SAU_AMD_POOL_MB", (size_t)64 << 10);
		static const size_t cap_pin = env_mb("SAU_AMD_PINNED_POOL_MB", (size_t)1 << 10);
		if (held + bytes > (pinned ? cap_pin : cap_dev)) return false;
		(pinned ? pin_ : dev_[dev & 15]).emplace(bytes, q);
		held += bytes;
		return true;
	}
	/* an allocation has failed: give the runtime back everything the pool holds for this device (or of page-locked memory) */
	size_t trim(bool pinned) {
		int dev = 0;
		(void)hipGetDevice(&dev);
		std::multimap<size_t, void *> out;
		{
			std::lock_guard<std::mutex> lk(mu_);
			out.swap(pinned ? pin_ : dev_[dev & 15]);
			for (auto &b : out) (pinned ? held_pin_ : held_dev_) -= b.first;
		}
		size_t n = 0;
		for (auto &b : out) { if (pinned) (void)hipHostFree(b.second); else (void)hipFree(b.second); n += b.first; }
		return n;
	}
private:
	static size_t env_mb(const char *name, size_t def_mb) {
		const char *v = getenv(name);
		return (v ? (size_t)atoll(v) : def_mb) << 20;
	}
	std::mutex mu_;
	std::multimap<size_t, void *> dev_[16], pin_;
	size_t held_dev_ = 0, held_pin_ = 0;
};

/* Streams too: creating one sets up a hardware queue, milliseconds on this runtime. */
class StreamPool {
public:
	static StreamPool &get() { static StreamPool *g = new StreamPool; return *g; }
	/* `kind` 1: the streams feedback chains run on (created with high priority: see chain_stream_) */
	hipStream_t take(int dev, int kind = 0) {
		std::lock_guard<std::mutex> lk(mu_);
		auto &v = free_[kind & 1][dev & 15];
		if (v.empty()) return nullptr;
		hipStream_t s = v.back();
		v.pop_back();
		return s;
	}
	void give(int dev, hipStream_t s, int kind = 0) {
		std::lock_guard<std::mutex> lk(mu_);
		auto &v = free_[kind & 1][dev & 15];
		if (v.size() < 8) v.push_back(s); else (void)hipStreamDestroy(s);
	}
private:
	std::mutex mu_;
	std::vector<hipStream_t> free_[2][16];
};

/* Launches of the single-pass running-sum build whose voices have waves in more than one workgroup wait for each
 * other's sums across workgroups. One such launch alone always finishes (its workgroups are dispatched in order and
 * a voice's lie next to each other, so what a blocked workgroup waits for is the next to get a CU), two at once
 * from different generators could hold each other's CUs: within the process they take turns, through an event
 * chain per device. (Launches whose voices each sit inside one workgroup need none of this.) */
class SpreadLaunchOrder {
public:
	static SpreadLaunchOrder &get() { static SpreadLaunchOrder *g = new SpreadLaunchOrder; return *g; }
	/* `launch` enqueues the kernel on `stream`; it runs after the device's previous such launch has finished */
	template <typename F> bool ordered(int dev, hipStream_t stream, F launch) {
		std::lock_guard<std::mutex> lk(mu_);
		Dev &d = dev_[dev & 15];
		if (d.last && hipStreamWaitEvent(stream, d.last, 0) != hipSuccess) return false;
		if (!launch()) return false;
		hipEvent_t &e = d.ring[d.next++ & 63];
		if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
		if (hipEventRecord(e, stream) != hipSuccess) return false;
		d.last = e; /* (a wait captures the event's state when it is enqueued: re-recording it 64 launches on is harmless) */
		return true;
	}
private:
	struct Dev { hipEvent_t ring[64] = {}; hipEvent_t last = nullptr; unsigned next = 0; };
	std::mutex mu_;
	Dev dev_[16];
};

static void *pool_alloc(bool pinned, size_t &bytes, std::string &err) {
	bytes = BufPool::bucket(bytes);
	void *q = BufPool::get().take(pinned, bytes);
	if (q) return q;
	hipError_t e = pinned ? hipHostMalloc(&q, bytes, hipHostMallocDefault) : hipMalloc(&q, bytes);
	if (e != hipSuccess) {
		/* The failed call's error stays the thread's "last error" (successful calls do not clear it on this runtime):
		 * consume it, or the hipGetLastError() checks behind later launches report this allocation against them -- a
		 * caller that has a way on without the block (the block loop in place of the chains' rows) would fail anyway.
		 * Then once more with the pool's idle blocks given back to the runtime. */
		(void)hipGetLastError();
		if (BufPool::get().trim(pinned)) {
			e = pinned ? hipHostMalloc(&q, bytes, hipHostMallocDefault) : hipMalloc(&q, bytes);
			if (e == hipSuccess) return q;
			(void)hipGetLastError();
		}
		err = std::string(pinned ? "hipHostMalloc: " : "hipMalloc: ") + hipGetErrorString(e);
		return nullptr;
	}
	return q;
}
static void pool_free(bool pinned, void *q, size_t bytes) {
	if (!q) return;
	if (!BufPool::get().give(pinned, q, bytes)) { if (pinned) (void)hipHostFree(q); else (void)hipFree(q); }
}

/* hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and kernel, remembered in static tables: generators
 * driven from different host threads share them, so check-then-raise runs under one lock (values only ever grow) */
static std::mutex &attr_mutex() { static std::mutex *m = new std::mutex; return *m; }
static bool raise_lds_attr(const void *fn, size_t lds, size_t &configured, std::string &err) {
	std::lock_guard<std::mutex> lk(attr_mutex());
	if (lds > configured) {
		hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) { err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); return false; }
		configured = lds;
	}
	return true;
}

template <typename T, bool PINNED> struct PoolBuf {
	T *p = nullptr;
	size_t cap = 0, bytes = 0;
	PoolBuf() = default;
	PoolBuf(const PoolBuf &) = delete;
	PoolBuf &operator=(const PoolBuf &) = delete;
	~PoolBuf() { release(); }
	/* growing replaces the block: the caller has drained the stream that used the old one */
	bool ensure(size_t n, std::string &err, bool keep = false) {
		if (n <= cap) return true;
		size_t nbytes = (n + n / 4 + 16) * sizeof(T);
		T *q = (T *)pool_alloc(PINNED, nbytes, err);
		if (!q) return false;
		if (keep && p && cap) {
			hipError_t e = PINNED ? (memcpy(q, p, cap * sizeof(T)), hipSuccess)
			                      : hipMemcpy(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice);
			if (e != hipSuccess) { err = hipGetErrorString(e); return false; }
		}
		if (p) { (void)hipDeviceSynchronize(); pool_free(PINNED, p, bytes); }
		p = q; bytes = nbytes; cap = nbytes / sizeof(T);
		return true;
	}
	void release() { pool_free(PINNED, p, bytes); p = nullptr; cap = 0; bytes = 0; }
};
template <typename T> using DevBuf = PoolBuf<T, false>;
template <typename T> using PinBuf = PoolBuf<T, true>; /* page-locked staging for async copies */

/* Hermite coefficient tables (sau/wave.h:127-141 evaluated per table entry) are a function of the
 * PILUTs alone: built and uploaded once per device and table set, shared by every generator. */
struct TableSet {
	int dev;
	std::vector<float> piluts;
	WaveConst wconst[12];
	HerpC23 *c23;
	HerpC01 *c01;
	WaveConst *wc;
};
static const TableSet *shared_tables(const float *piluts, const WaveConst *wconst, std::string &err) {
	static std::mutex mu;
	static std::vector<TableSet *> sets;
	int dev = 0;
	(void)hipGetDevice(&dev);
	std::lock_guard<std::mutex> lk(mu);
	for (const TableSet *t : sets)
		if (t->dev == dev && !memcmp(t->piluts.data(), piluts, (size_t)12 * WAVE_LEN * sizeof(float)) &&
		    !memcmp(t->wconst, wconst, sizeof t->wconst))
			return t;
	std::vector<HerpC23> h23((size_t)12 * WAVE_LEN);
	std::vector<HerpC01> h01((size_t)12 * WAVE_LEN);
	for (uint32_t wv = 0; wv < 12; ++wv)
		for (uint32_t i = 0; i < WAVE_LEN; ++i) {
			if (!herp_c1_scalable(piluts + (size_t)wv * WAVE_LEN, i)) {
				err = "wave table has slopes below 2^-100: unsupported";
				return nullptr;
			}
			herp_coeffs(piluts + (size_t)wv * WAVE_LEN, i,
					h23[(size_t)wv * WAVE_LEN + i], h01[(size_t)wv * WAVE_LEN + i]);
		}
	TableSet *t = new TableSet;
	t->dev = dev;
	t->piluts.assign(piluts, piluts + (size_t)12 * WAVE_LEN);
	memcpy(t->wconst, wconst, sizeof t->wconst);
	t->c23 = nullptr; t->c01 = nullptr; t->wc = nullptr;
	hipError_t e = hipMalloc((void **)&t->c23, h23.size() * sizeof(HerpC23));
	if (e == hipSuccess) e = hipMalloc((void **)&t->c01, h01.size() * sizeof(HerpC01));
	if (e == hipSuccess) e = hipMalloc((void **)&t->wc, 12 * sizeof(WaveConst));
	if (e == hipSuccess) e = hipMemcpy(t->c23, h23.data(), h23.size() * sizeof(HerpC23), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(t->c01, h01.data(), h01.size() * sizeof(HerpC01), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(t->wc, wconst, 12 * sizeof(WaveConst), hipMemcpyHostToDevice);
	if (e != hipSuccess) {
		err = std::string("wave tables: ") + hipGetErrorString(e);
		(void)hipFree(t->c23); (void)hipFree(t->c01); (void)hipFree(t->wc);
		delete t;
		return nullptr;
	}
	sets.push_back(t); /* generators keep plain pointers: sets live as long as the process (0.6 MB each) */
	return t;
}

int device_count() {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

bool device_pci_bus_id(int dev, char *buf, int len) {
	if (!buf || len < 16) return false;
	buf[0] = 0;
	return hipDeviceGetPCIBusId(buf, len, dev) == hipSuccess;
}

class HipBackendImpl : public HipBackend {
public:
	explicit HipBackendImpl(int device = -1) : want_dev_(device) {}
	/* every entry point runs on this backend's device, whatever the host thread's current one is
	 * (another generator on another GPU, torch.cuda.set_device): allocations, the pools (keyed by
	 * the current device) and launches all follow it */
	void use_device() { (void)hipSetDevice(dev_); }

	~HipBackendImpl() override {
		use_device();
		/* the buffers go back to the pool (member destructors): nothing may still be using them */
		if (stream_) (void)hipStreamSynchronize(stream_);
		if (chain_stream_) { (void)hipStreamSynchronize(chain_stream_); StreamPool::get().give(dev_, chain_stream_, 1); }
		for (hipEvent_t e : chain_ev_) (void)hipEventDestroy(e);
		for (int i = 0; i < 4; ++i) if (fetch_ev_[i]) (void)hipEventDestroy(fetch_ev_[i]);
		if (after_ev_) (void)hipEventDestroy(after_ev_);
		for (auto &e : events_) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
		if (stream_) StreamPool::get().give(dev_, stream_); /* drained above */
	}

	bool init(const BackendConfig &cfg, std::string &err) override {
		cfg_ = cfg;
		int dev = 0;
		const char *env = getenv("SAU_AMD_DEVICE");
		if (want_dev_ >= 0) dev = want_dev_; /* (sauAmd_create_Batch_on: the caller's choice goes before the environment's) */
		else if (env) dev = atoi(env);
		if (dev < 0 || dev >= device_count()) { err = "no HIP device " + std::to_string(dev) + " (" + std::to_string(device_count()) + " visible)"; return false; }
		HIP_OK(hipSetDevice(dev));
		dev_ = dev;
		stream_ = StreamPool::get().take(dev);
		if (!stream_) HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
		static std::mutex prop_mu;
		static size_t dev_lds[16]; /* per device: hipGetDeviceProperties costs a millisecond */
		static uint32_t dev_cus[16];
		static size_t dev_free[16]; /* bytes free when the process first opened the device: the chain rows' budget follows it */
		{
			std::lock_guard<std::mutex> lk(prop_mu);
			if (!dev_lds[dev & 15]) {
				hipDeviceProp_t prop;
				HIP_OK(hipGetDeviceProperties(&prop, dev));
				dev_lds[dev & 15] = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor
				                                                          : prop.sharedMemPerBlock;
				dev_cus[dev & 15] = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : FK_GRID;
				size_t mfree = 0, mtotal = 0; /* the budget of the feedback chains' rows follows what this device has free */
				if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) dev_free[dev & 15] = mfree;
			}
			free_hint_ = dev_free[dev & 15];
			lds_limit_ = dev_lds[dev & 15];
			/* the time-parallel kernels' grids are one 1024-thread workgroup per CU at most: waves that wait for other
			 * workgroups' sums (spread look-back launches) need the whole grid resident, so the grid follows the CUs
			 * this device really has (a partitioned or CU-masked device has fewer than 256) */
			fk_grid_ = dev_cus[dev & 15] < FK_GRID ? dev_cus[dev & 15] : FK_GRID;
			/* the per-XCD task queues and the launch's own mixing take chunk k mod 8 for XCD k mod 8 and hand rows from wave to wave
			 * through ONE XCD's L2 (k_fast_types.h): only on a device that is the whole MI355X -- 256 CUs = 8 XCDs x 32. A partition
			 * (CPX: 32 CUs, one XCD) or a masked device keeps the single counter and mix_kernel */
			eight_xcds_ = dev_cus[dev & 15] == 256;
			cus_ = dev_cus[dev & 15];
			if (const char *fg = tune_env("SAU_AMD_FK_GRID")) { const int n = atoi(fg); if (n >= 1 && n <= (int)FK_GRID) fk_grid_ = (uint32_t)n; }
		}
		if (lds_limit_ > 160 * 1024) lds_limit_ = 160 * 1024;
		if (const char *ll = tune_env("SAU_AMD_LDS_LIMIT")) lds_limit_ = (size_t)atol(ll);
		const char *wt = tune_env("SAU_AMD_GEOMETRY"); /* "4x4" (default) or "8x2" */
		geo_ = (wt && !strcmp(wt, "8x2")) ? 0 : 1; /* default 4 waves x 4 samples per lane */
		debug_ = getenv("SAU_AMD_DEBUG") != nullptr;
		fast_enabled_ = tune_env("SAU_AMD_NO_FAST") == nullptr;
		seq_enabled_ = tune_env("SAU_AMD_NO_SEQ") == nullptr; /* running-sum phases in the time-parallel kernel */
		chain_enabled_ = tune_env("SAU_AMD_NO_CHAIN") == nullptr; /* feedback recurrences with lanes = voices */
		chain_inline_ = tune_env("SAU_AMD_NO_CHAIN_INLINE") == nullptr; /* chains fed from their own lines: chain_kernel's feeder waves evaluate them */
		chain_early_ = tune_env("SAU_AMD_NO_EARLY_CHAINS") == nullptr;
		inc_rows_enabled_ = tune_env("SAU_AMD_NO_INC_ROWS") == nullptr;
		lookback_enabled_ = tune_env("SAU_AMD_NO_LOOKBACK") == nullptr; /* single-pass running sums */
		if (const char *lr = tune_env("SAU_AMD_LOOK_ROWS")) look_rows_ = (uint32_t)atoi(lr);
		if (const char *lm = tune_env("SAU_AMD_LOOK_MIN_VOICES")) look_min_voices_ = (uint32_t)atoi(lm);
		if (const char *cc = tune_env("SAU_AMD_CHAIN_CHUNKS")) { /* pipeline depth of a segment with chains (1: off; default: by frames) */
			const int n = atoi(cc);
			chain_chunks_ = n >= 32 ? 32 : n >= 1 ? (uint32_t)n : 1;
		}
		if (const char *cf = tune_env("SAU_AMD_CHAIN_CHUNK_FRAMES")) { const int n = atoi(cf); if (n >= 4096) chain_chunk_frames_ = (uint32_t)n; }
		two_pass_enabled_ = tune_env("SAU_AMD_NO_TWO_PASS") == nullptr; /* ... in two passes where possible */
		if (const char *lr = tune_env("SAU_AMD_LEAN_ROWS")) lean_rows_ = (uint32_t)atoi(lr);
		mix_few_enabled_ = tune_env("SAU_AMD_NO_MIX_FEW") == nullptr;
		early_mix_enabled_ = tune_env("SAU_AMD_NO_EARLY_MIX") == nullptr;
		inmix_enabled_ = tune_env("SAU_AMD_NO_INMIX") == nullptr;
		tailmix_enabled_ = tune_env("SAU_AMD_TAILMIX") != nullptr;
		duo_enabled_ = tune_env("SAU_AMD_NO_DUO") == nullptr;
		mix64_enabled_ = tune_env("SAU_AMD_NO_MIX64") == nullptr;
		xcd_queues_ = tune_env("SAU_AMD_NO_XCD_QUEUES") == nullptr;
		inmix_taper_ = tune_env("SAU_AMD_INMIX_TAPER") != nullptr;
		inmix_report_ = tune_env("SAU_AMD_INMIX_REPORT") != nullptr;
		if (const char *ia = tune_env("SAU_AMD_INMIX_AT")) inmix_at_ = (uint32_t)atoi(ia) & 15u;
		if (const char *mv = tune_env("SAU_AMD_INMIX_MIN_VOICES")) inmix_min_voices_ = (uint32_t)atoi(mv);
		if (const char *ms = tune_env("SAU_AMD_INMIX_MIN_STEPS")) inmix_min_steps_ = (uint32_t)atoi(ms);
		short_last_chunk_ = tune_env("SAU_AMD_NO_SHORT_LAST_CHUNK") == nullptr;
		lean_enabled_ = tune_env("SAU_AMD_NO_LEAN") == nullptr; /* chains' passes in a build without the several-pass sums */
		dyn_enabled_ = tune_env("SAU_AMD_NO_DYN") == nullptr; /* closed-form launches deal tasks out through a counter */
		wide_tabs_ = tune_env("SAU_AMD_NO_WIDE_TABS") == nullptr; /* closed-form launches with f64 [c1, c0] table entries in LDS */
		if (const char *mr = tune_env("SAU_AMD_MORE_ROWS")) more_rows_ = (uint32_t)atoi(mr); /* 0, 10 or 12 */
		if (const char *dg = tune_env("SAU_AMD_DYN_GROUPS")) { const int n = atoi(dg); dyn_groups_ = n >= 1 ? (uint32_t)n : 1u; }
		if (const char *ca = tune_env("SAU_AMD_CHAIN_ALONE")) chain_alone_ = atoi(ca) != 0;
		inner_enabled_ = tune_env("SAU_AMD_NO_INNER") == nullptr;
		if (const char *df = tune_env("SAU_AMD_DYN_FLOOR")) { const int n = atoi(df); dyn_floor_ = n >= 1 ? (uint32_t)n : 1u; }
		if (const char *dt = tune_env("SAU_AMD_DYN_MIN_TASKS")) { const int n = atoi(dt); dyn_min_tasks_ = n >= 1 ? (uint32_t)n : 1u; }
		/* voices per segment from which feedback voices get sixteen one-wave teams per workgroup
		 * (0: never; 1: always, also without feedback -- tests) */
		multi_min_ = 256;
		if (const char *fr = tune_env("SAU_AMD_FAST_ROWS")) { /* 8 (default), 4 or 2 */
			const int r = atoi(fr);
			fast_rows_ = r >= 8 ? 8 : r >= 6 ? 6 : r >= 5 ? 5 : r >= 4 ? 4 : 2;
		}
		if (const char *mm = tune_env("SAU_AMD_MULTI_MIN")) multi_min_ = (uint32_t)atol(mm);
		if (const char *mt = tune_env("SAU_AMD_MULTI_TEAMS")) multi_teams_ = atoi(mt) == 8 ? 8u : 16u;
		if (!ops_.ensure(cfg.op_count ? cfg.op_count : 1, err)) return false;
		HIP_OK(hipMemsetAsync(ops_.p, 0, ops_.cap * sizeof(DevOp), stream_));
		tables_ = shared_tables(cfg.piluts, cfg.wconst, err);
		if (!tables_) return false;
		memcpy(wconst_, cfg.wconst, sizeof wconst_);
		return true;
	}

	bool reserve_frames(uint32_t max_frames, bool stereo, std::string &err) override {
		use_device();
		HIP_OK(hipStreamSynchronize(stream_));
		row_stride_ = (max_frames + 63) & ~63u;
		pcm_row_ = (size_t)row_stride_ * 2; /* room for stereo */
		(void)stereo;
		{
			if (!pcm_.ensure(pcm_row_ * cfg_.n_streams, err)) return false;
			/* (not cleared: every frame of a run is written by a mixer or cleared by zero_pcm, and no frame behind a run's
			 * length is ever handed out -- until round 6 a memset here, 32 us for a config-4 batch's 677 MB) */
		}
		set_.vout_rows = 0; /* re-sized on the next render */
		return true;
	}

	/* Host data for the device goes through a page-locked arena and asynchronous copies on the
	 * generator's stream: the copies are ordered behind the kernels still reading the old
	 * contents, and the host goes on to prepare the next segment instead of waiting (a script
	 * with many events is a chain of short segments). The arena is reused from its start once
	 * the stream has drained. */
	void *stage(const void *src, size_t bytes, std::string &err) {
		const size_t need = (bytes + 63) & ~(size_t)63;
		if (arena_used_ + need > arena_.cap) {
			hipError_t e = hipStreamSynchronize(stream_);
			if (e != hipSuccess) { err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); return nullptr; }
			arena_used_ = 0;
			const size_t want = need > ((size_t)4 << 20) ? need : ((size_t)4 << 20);
			if (want > arena_.cap && !arena_.ensure(want, err)) return nullptr;
		}
		void *p = arena_.p + arena_used_;
		arena_used_ += need;
		memcpy(p, src, bytes);
		return p;
	}
	bool send(void *dst, const void *src, size_t bytes, std::string &err) {
		if (!bytes) return true;
		void *h = stage(src, bytes, err);
		if (!h) return false;
		HIP_OK(hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, stream_));
		return true;
	}

	bool upload_plans(const Step *steps, const FastIds *fast_ids, size_t n_steps, const uint32_t *op_ids,
			size_t n_ids, std::string &err) override {
		use_device();
		if (!steps_.ensure(n_steps ? n_steps : 1, err) || !op_ids_.ensure(n_ids ? n_ids : 1, err) ||
		    !fast_ids_.ensure(n_steps ? 2 * n_steps : 1, err))
			return false;
		n_steps_total_ = (uint32_t)n_steps;
		return send(steps_.p, steps, n_steps * sizeof(Step), err) &&
			send(fast_ids_.p, fast_ids, 2 * n_steps * sizeof(FastIds), err) &&
			send(op_ids_.p, op_ids, n_ids * sizeof(uint32_t), err);
	}

	bool apply_updates(const OpUpdate *recs, size_t n, std::string &err) override {
		use_device();
		if (!n) return true;
		if (!recs_.ensure(n, err) || !send(recs_.p, recs, n * sizeof(OpUpdate), err)) return false;
		hipLaunchKernelGGL(event_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream_,
				ops_.p, recs_.p, (uint32_t)n, tables_->wc);
		HIP_OK(hipGetLastError());
		return true;
	}

	bool zero_pcm(uint32_t first_stream, uint32_t n_streams, uint32_t first_frame, uint32_t n_frames, bool stereo,
			std::string &err) override {
		use_device();
		if (!pcm_.p || !n_streams || !n_frames) return true;
		const size_t ch = stereo ? 2 : 1;
		int16_t *at = pcm_.p + pcm_row_ * first_stream + (size_t)first_frame * ch;
		if (n_streams == 1) HIP_OK(hipMemsetAsync(at, 0, (size_t)n_frames * ch * sizeof(int16_t), stream_));
		else HIP_OK(hipMemset2DAsync(at, pcm_row_ * sizeof(int16_t), 0, (size_t)n_frames * ch * sizeof(int16_t), n_streams, stream_));
		return true;
	}

	template <int W, int T, int V, bool HB = false>
	bool launch_render(const RenderParams &rp, uint32_t grid, size_t lds, std::string &err) {
		static size_t configured[16]; /* per device (function attributes are per device) */
		if (!raise_lds_attr((const void *)render_kernel<W, T, V, HB>, lds, configured[dev_ & 15], err)) return false;
		hipLaunchKernelGGL((render_kernel<W, T, V, HB>), dim3(grid), dim3(64 * W * V), lds, stream_, rp);
		HIP_OK(hipGetLastError());
		return true;
	}

	bool render(const SegmentDesc &seg, std::string &err) override {
		use_device();
		if (!seg.n_voices) return true;
		++acc_launches_; /* segments rendered */
		early_mixed_blocks_ = 0;
		inmix_live_ = false;
		const size_t tab_bytes = (size_t)WAVE_LEN * (sizeof(HerpC23) + sizeof(HerpC01));
		/* (the time-parallel kernels keep their LDS copy of a table in another form: FAST_TAB_BYTES, k_fast_types.h) */
		const size_t ftab_bytes = FAST_TAB_BYTES;
		/* Block-loop geometry. Few voices: W waves share one voice (4x4 or 8x2
		 * frames per lane). Many voices: sixteen single-wave teams per workgroup,
		 * each with its own voice, so that every CU has 16 voices in flight. */
		uint32_t W = geo_ ? 4 : 8, T = geo_ ? 4 : 2, V = 1;
		bool HB = false; /* block buffers in HBM (render_kernel<1, 1, 1, true>) */
		auto team_size = [&](uint32_t w, uint32_t t) {
			size_t b = HB ? sizeof(Misc) + 64 /* (buffers, operator records and steps in HBM) */
				: (size_t)w * 64 * t * sizeof(float) * seg.n_slots + (size_t)seg.max_ops * sizeof(DevOp) +
				sizeof(Misc) + (size_t)seg.max_steps * sizeof(Step) + 64;
			return (b + 15) & ~(size_t)15;
		};
		if (multi_min_ && seg.n_voices >= multi_min_ && (seg.serial || multi_min_ == 1)) {
			const size_t need_tab = seg.wave_mask ? tab_bytes : 0;
			/* longer blocks amortise the per-step bookkeeping: 255 samples per block measured
			 * 14 % faster than 127 on BASELINE config 5 */
			const uint32_t mt = multi_teams_; /* single-wave teams per workgroup: 8 (512 threads: 256 VGPRs each, no scratch) or 16 */
			if (mt * team_size(1, 4) + need_tab <= lds_limit_) { W = 1; T = 4; V = mt; }
			else if (mt * team_size(1, 3) + need_tab <= lds_limit_) { W = 1; T = 3; V = mt; }
			else if (mt * team_size(1, 2) + need_tab <= lds_limit_) { W = 1; T = 2; V = mt; }
			else if (mt * team_size(1, 1) + need_tab <= lds_limit_) { W = 1; T = 1; V = mt; }
		}
		/* a voice with very many block buffers: one wave, one frame per lane (256 B per buffer) */
		if (V == 1 && team_size(W, T) > lds_limit_) { W = 1; T = 1; }
		/* more of them than LDS holds (wide plans: graphs nested hundreds of levels deep): the buffers go to HBM */
		if (V == 1 && W == 1 && team_size(1, 1) > lds_limit_) HB = true;
		/* LDS budget: slots + operator cache + misc per team, rest for tables */
		const size_t team_bytes = team_size(W, T);
		size_t fixed = team_bytes * V;
		if (fixed > lds_limit_) {
			err = "voice too large for one workgroup's LDS (block buffers + operator states)";
			return false;
		}
		RenderParams rp;
		memset(&rp, 0, sizeof rp);
		rp.team_bytes = (uint32_t)team_bytes;
		uint32_t n_tabs = 0;
		/* keep two workgroups per CU when possible: cap at half the LDS */
		size_t budget = lds_limit_ / 2 > fixed ? lds_limit_ / 2 - fixed : 0;
		if (V > 1) budget = lds_limit_ - fixed;
		if (budget < tab_bytes && lds_limit_ - fixed >= tab_bytes) budget = tab_bytes;
		for (int wv = 0; wv < 12; ++wv) {
			rp.tab_of_wave[wv] = -1;
			if ((seg.wave_mask >> wv) & 1) {
				if ((n_tabs + 1) * tab_bytes <= budget) {
					rp.tab_of_wave[wv] = (int8_t)n_tabs;
					rp.wave_of_tab[n_tabs] = (uint8_t)wv;
					++n_tabs;
				}
			}
		}
		const size_t lds = n_tabs * tab_bytes + fixed;
		/* device buffers */
		MixSet &S = set_;
		if (!voices_.ensure(seg.n_voices, err) || !S.vinfo.ensure(seg.n_voices, err)) return false;
		if (seg.n_voices > S.vout_rows || !S.vout.p) {
			HIP_OK(hipStreamSynchronize(stream_));
			if (!S.vout.ensure((size_t)seg.n_voices * row_stride_, err)) return false;
			S.vout_rows = seg.n_voices;
		}
		if (seg.n_pan_rows && !S.pan.ensure((size_t)seg.n_pan_rows * row_stride_, err)) return false;
		if (!S.mstreams.ensure(seg.n_streams, err)) return false;
		/* descriptors go through page-locked staging that is reused once the
		 * previous segment's copies have left it */
		/* in steady state (no event between two segments) the descriptors repeat: upload only changes */
		ms_host_.resize(seg.n_streams);
		uint32_t max_write = 0, max_rows = 0;
		for (uint32_t s = 0; s < seg.n_streams; ++s) {
			MixStream &m = ms_host_[s];
			memset(&m, 0, sizeof m);
			m.first_row = seg.streams[s].first_voice;
			m.n_rows = seg.streams[s].n_voices;
			m.amp_scale = seg.streams[s].amp_scale;
			m.write_len = seg.streams[s].write_len;
			m.pcm = pcm_.p + pcm_row_ * s;
			if (m.write_len > max_write) max_write = m.write_len;
			if (m.n_rows > max_rows) max_rows = m.n_rows;
		}
		const bool same_voices = voices_sent_.size() == seg.n_voices && voices_dev_ == voices_.p &&
			memcmp(voices_sent_.data(), seg.voices, seg.n_voices * sizeof(VoiceDesc)) == 0;
		const bool same_ms = S.ms_sent.size() == seg.n_streams && S.ms_dev == S.mstreams.p &&
			memcmp(S.ms_sent.data(), ms_host_.data(), seg.n_streams * sizeof(MixStream)) == 0;
		if (!same_voices) {
			if (!send(voices_.p, seg.voices, seg.n_voices * sizeof(VoiceDesc), err)) return false;
			voices_sent_.assign(seg.voices, seg.voices + seg.n_voices);
			voices_dev_ = voices_.p;
		}
		if (!same_ms) {
			if (!send(S.mstreams.p, ms_host_.data(), seg.n_streams * sizeof(MixStream), err)) return false;
			S.ms_sent = ms_host_;
			S.ms_dev = S.mstreams.p;
		}
		rp.voices = voices_.p; rp.steps = steps_.p; rp.op_ids = op_ids_.p; rp.ops = ops_.p;
		rp.vout = S.vout.p; rp.pan = S.pan.p; rp.vinfo = S.vinfo.p;
		rp.g_c23 = tables_->c23; rp.g_c01 = tables_->c01;
		rp.row_stride = row_stride_; rp.seg_len = seg.len;
		rp.n_slots = seg.n_slots; rp.max_ops = seg.max_ops; rp.n_tabs = n_tabs;
		rp.max_steps = seg.max_steps; rp.n_main = seg.n_main;
		rp.fast_done = nullptr;
		memcpy(rp.wc, wconst_, sizeof wconst_);
		/* ---- time-parallel path first; the block loop continues after it ---- */
		{
			const uint32_t fmax_steps = seg.max_steps < 64 ? seg.max_steps : 64;
			/* rows per wave and pass: as many as LDS holds beside one wave table
			 * (more rows amortise the per-step work: 8 rows measured 8 % faster
			 * than 4, 4 rows 28 % faster than 2) */
			uint32_t FT = fast_rows_;
			auto fewer = [](uint32_t t) { return t > 6 ? 6u : t > 5 ? 5u : t > 4 ? 4u : 2u; }; /* 8, 6, 5, 4 or 2 rows per pass */
			/* The build with all the running-sum code needs more registers: 8 rows per pass would spill. Where the
			 * single-pass (look-back) build serves, it takes those voices and the closed-form ones at the full rows
			 * per pass, and the full build's launches only see what is left (voices with feedback chains, voices
			 * one wave walks in order). */
			const bool look_split = seq_enabled_ && seg.may_scan && two_pass_enabled_ && lookback_enabled_ && look_rows_ != 0 &&
				seg.n_voices >= look_min_voices_;
			if (seq_enabled_ && seg.may_scan && FT > 4 && !look_split) FT = 4;
			if (look_split && FT > look_rows_) FT = look_rows_ >= 8 ? 8 : look_rows_ >= 6 ? 6 : look_rows_ >= 5 ? 5 : look_rows_ >= 4 ? 4 : 2;
			/* block buffers: without frequency blocks, or with them when some voice may need
			 * the sequential scan (ramped or modulated frequencies) */
			const bool seq_ok = seq_enabled_ && seg.may_scan;
			const uint32_t n_fast = seq_ok && seg.n_fast_full > seg.n_fast ? seg.n_fast_full : seg.n_fast;
			auto area_of = [&](uint32_t t) {
				return (size_t)n_fast * 64 * t * sizeof(float) + (size_t)fmax_steps * sizeof(unsigned long long);
			};
			const size_t one_tab = seg.wave_mask ? ftab_bytes : 0;
			const size_t look_lds = look_split ? LOOK_LDS_BYTES : 0;
			while (FT > 2 && 16 * area_of(FT) + one_tab + look_lds + 1024 > lds_limit_) FT = fewer(FT);
			{ /* every wave table the segment uses in LDS is worth more than rows per pass (an oscillator whose table
			   * is left out reads it from L2 per sample): fewer rows where that makes them all fit */
				const size_t need = (size_t)__builtin_popcount(seg.wave_mask) * ftab_bytes + look_lds + 1024;
				uint32_t t = FT;
				while (t > 4 && 16 * area_of(t) + need > lds_limit_) t = fewer(t); /* (but not below 4 rows: that costs more) */
				if (16 * area_of(t) + need <= lds_limit_) FT = t;
			}
			size_t area = area_of(FT);
			const bool use_fast = fast_enabled_ && (16 * area + look_lds + 1024 <= lds_limit_);
			const uint32_t FTM = FT > 4 && seq_enabled_ && seg.may_scan ? 4 : FT; /* rows per pass of the full build */
			if (!finfo_.ensure(seg.n_voices, err) || !fdone_.ensure(seg.n_voices, err) ||
			    !worklist_.ensure(seg.n_voices, err) || !work_count_.ensure(4, err) ||
			    !fsteps_.ensure((size_t)FAST_LISTS * seg.n_voices * fmax_steps * sizeof(FastStep), err) ||
			    !flines_.ensure((size_t)FAST_LISTS * seg.n_voices * fmax_steps * sizeof(FastLine), err) ||
			    !faux_.ensure((size_t)FAST_LISTS * seg.n_voices * fmax_steps * sizeof(FastAux), err)) return false;
			FastParams fp;
			memset(&fp, 0, sizeof fp);
			fp.voices = voices_.p; fp.steps = steps_.p; fp.fast_ids = fast_ids_.p; fp.op_ids = op_ids_.p; fp.ops = ops_.p;
			fp.vout = S.vout.p; fp.pan = S.pan.p; fp.info = finfo_.p; fp.fast_done = fdone_.p;
			fp.worklist = worklist_.p; fp.work_count = work_count_.p; fp.vinfo = S.vinfo.p;
			fp.g_c23 = tables_->c23; fp.g_c01 = tables_->c01; fp.fsteps = (FastStep *)fsteps_.p; fp.flines = (FastLine *)flines_.p; fp.faux = (FastAux *)faux_.p;
			fp.row_stride = row_stride_; fp.n_voices = seg.n_voices; fp.n_fast = n_fast;
			fp.seq_enable = seq_ok ? 1u : 0u; fp.ids_full_ofs = n_steps_total_;
			fp.scan = nullptr; fp.scan_groups = 0; fp.mode = 0;
			if (xcd_queues_ && eight_xcds_) { /* the closed-form launch's task queues, one per XCD (k_fast_types.h) */
				if (!inmix_ctl_.ensure(INMIX_WORDS, err)) return false;
				fp.inmix = inmix_ctl_.p;
			}
			if (seq_ok && two_pass_enabled_) {
				/* row groups per voice at most: rows hold at least 32 new frames (H <= 32) */
				fp.scan_groups = seg.len / (32 * FTM) + 2;
				/* (the several-pass form's sums: with look-back only voices that have, or may get, feedback chains take it) */
				if (!look_split || seg.n_chain_rows) {
					if (!scan_.ensure((size_t)seg.n_voices * FAST_MAX_SCAN * fp.scan_groups, err)) return false;
					fp.scan = scan_.p;
				}
				if (look_split && seg.n_look_rows) {
					/* Waves per voice of the single-pass build's launch (as many as the voice has row groups, up to 64,
					 * when voices are few). Voices whose waves sit in one workgroup (1, 2, 4, 8 or 16 of them) look back
					 * through rings in LDS and need no words in HBM; only a launch that spreads voices over
					 * neighbouring workgroups does. Those words (8 B x 2 x row groups per running-sum oscillator) are
					 * capped at 1 GiB: beyond that the launch keeps every voice inside a workgroup instead. */
					const uint32_t groups = (seg.len + (60 * FT) - 1) / (60 * FT);
					unsigned long long wpv = ((unsigned long long)fk_grid_ * 16) / seg.n_voices;
					if (wpv > 64) wpv = 64;
					if (wpv > groups) wpv = groups;
					if (wpv < 1) wpv = 1;
					const bool no_lds = tune_env("SAU_AMD_LOOK_NO_LDS") != nullptr;
					auto is_inside = [&](unsigned long long w) { return w <= 16 && (16 % w) == 0 && !no_lds; };
					const size_t look_words = (size_t)seg.n_look_rows * 2 * fp.scan_groups;
					/* the words exist whenever they fit 1 GiB: the kernel then picks the waves per voice from the number of
					 * look-back voices analyze_kernel finds (the host only knows how many there may be) */
					look_words_real_ = look_words * sizeof(unsigned long long) <= ((size_t)1 << 30);
					if (!look_words_real_ && !is_inside(wpv) && !no_lds)
						wpv = wpv > 16 ? 16 : wpv > 8 ? 8 : wpv > 4 ? 4 : 2;
					look_wpv_ = (uint32_t)wpv;
					look_inside_ = !look_words_real_ && is_inside(wpv);
					if (look_words_real_ || !look_inside_) {
						/* look-back words: valid for this segment's epoch only, so a fresh block starts out zeroed */
						const unsigned long long *before = look_.p;
						if (!look_.ensure(look_words, err)) return false;
						if (look_.p != before || look_epoch_ >= (1u << 30) - 1) {
							HIP_OK(hipMemsetAsync(look_.p, 0, look_.cap * sizeof(unsigned long long), stream_));
							look_epoch_ = 0;
						}
						fp.look_epoch = ++look_epoch_;
					} else if (!look_.p && !look_.ensure(64, err)) {
						return false; /* (a token block: the kernel forms row addresses from it and never reads them) */
					}
					fp.look = look_.p;
				}
			}
			if (!pass_flags_.p) {
				if (!pass_flags_.ensure(FAST_FLAGS, err)) return false;
				HIP_OK(hipMemsetAsync(pass_flags_.p, 0, pass_flags_.cap * sizeof(uint32_t), stream_));
			}
			fp.pass_flags = pass_flags_.p;
			fp.sum_levels = seg.sum_levels >= FAST_MAX_LEVELS ? FAST_MAX_LEVELS : 2u;
			if (!repair_.ensure((size_t)seg.n_voices * FAST_REPAIR_WORDS, err)) return false;
			fp.repair = repair_.p;
			fp.repair_on = tune_env("SAU_AMD_NO_REPAIR") ? 0u : 1u;
			fp.max_ops = seg.max_ops; fp.max_steps = fmax_steps; fp.np = 64; fp.rows = FT; fp.rows_multi = FTM;
			fp.enable = use_fast ? 1u : 0u;
			/* saved phase increments of running-sum oscillators (sum pass -> final pass), one segment long */
			/* (only voices that take several passes use them -- with look-back those with feedback chains -- and a
			 * bank of thousands of those would ask for gigabytes: beyond 1 GiB the final pass recomputes instead) */
			if (inc_rows_enabled_ && use_fast && fp.scan && seg.n_inc_rows && seg.len <= sauengine::CHAIN_SEG &&
			    (size_t)seg.n_inc_rows * 2 * ((seg.len + 63) & ~63u) * sizeof(uint32_t) <= ((size_t)1 << 30)) {
				const uint32_t istride = (seg.len + 63) & ~63u;
				if (!inc_rows_.ensure((size_t)seg.n_inc_rows * 2 * istride + 64, err)) return false;
				fp.inc_rows = inc_rows_.p; fp.inc_stride = istride; fp.n_inc_rows = seg.n_inc_rows;
			}
			/* feedback chains: a pair of rows per chain in HBM, one segment long (the engine keeps segments
			 * with such voices within CHAIN_SEG frames); without them those voices take the block loop */
			bool chains = chain_enabled_ && use_fast && fp.scan && seg.serial && seg.n_chain_rows &&
				seg.len <= sauengine::chain_seg_frames(seg.n_chain_rows, chain_rows_budget());
			if (chains) {
				const uint32_t cstride = (seg.len + 63) & ~63u;
				/* The rows are the one large allocation that depends on how long the engine cut the segment. When the device
				 * cannot give them (a smaller or fuller GPU than the budget assumed, several engines on it) this segment's
				 * feedback voices take the block loop -- slow, exact -- and later segments are cut at CHAIN_SEG again
				 * (ADVICE r03). SAU_AMD_CHAIN_ROWS_FAIL (tests) makes the first such allocation of an engine fail -- by not
				 * asking, or ("real") by asking hipMalloc for 64 TiB, which fails the way a full device does and
				 * leaves its error behind as the thread's last one. */
				std::string rows_err;
				const char *rft = chain_rows_failed_once_ ? nullptr : tune_env("SAU_AMD_CHAIN_ROWS_FAIL");
				const bool rows_fail_test = rft && strcmp(rft, "real") != 0;
				const size_t rows_want = (rft && !rows_fail_test) ? (size_t)1 << 44 : (size_t)seg.n_chain_rows * 2 * cstride + 64;
				if (rows_fail_test || !chain_rows_.ensure(rows_want, rows_err)) {
					chain_rows_failed_once_ = true;
					++chain_rows_failures_; /* (this backend's budget halves: Backend::chain_rows_budget) */
					if (debug_) fprintf(stderr, "saugns_amd: %zu bytes of chain rows not available (%s): block loop for this segment\n",
							((size_t)seg.n_chain_rows * 2 * cstride + 64) * sizeof(float), rows_err.c_str());
					chains = false;
				}
			}
			if (chains) {
				const uint32_t cstride = (seg.len + 63) & ~63u;
				if (!chain_desc_.ensure(seg.n_chain_slots, err) ||
				    !fplines_.ensure((size_t)FAST_LISTS * seg.n_voices * fmax_steps * sizeof(FastLine), err)) return false;
				/* (lanes that belong to no voice -- the engine begins every kind of R feedback chain on a wave of its own -- must read
				 * as unused: analyze_kernel clears the descriptors of the voices' lanes only) */
				if (seg.chain_rows_padded) HIP_OK(hipMemsetAsync(chain_desc_.p, 0, (size_t)seg.n_chain_slots * sizeof(ChainDesc), stream_));
				fp.chain_rows = chain_rows_.p; fp.chain_stride = cstride; fp.n_chain_rows = seg.n_chain_rows; fp.n_chain_slots = seg.n_chain_slots;
				fp.chain_desc = chain_desc_.p; fp.fplines = (FastLine *)fplines_.p;
				fp.chain_inline = chain_inline_ ? 1u : 0u;
				fp.chain_early_ok = chain_early_ ? 1u : 0u;
				uint32_t ct = 0;
				for (int wv = 0; wv < 12; ++wv) {
					fp.ctab_of_wave[wv] = -1;
					if (((seg.wave_mask >> wv) & 1) && (ct + 1) * (size_t)CHAIN_TAB_BYTES + CHAIN_IO_BYTES + 1024 <= lds_limit_) {
						fp.ctab_of_wave[wv] = (int8_t)ct;
						fp.cwave_of_tab[ct] = (uint8_t)wv;
						++ct;
					}
				}
				fp.n_ctabs = ct;
			}
			memcpy(fp.wc, wconst_, sizeof wconst_);
			uint32_t ft = 0;
			for (int wv = 0; wv < 12; ++wv) {
				fp.tab_of_wave[wv] = -1;
				if (use_fast && ((seg.wave_mask >> wv) & 1) &&
				    16 * area + look_lds + (ft + 1) * ftab_bytes + 1024 <= lds_limit_) {
					fp.tab_of_wave[wv] = (int8_t)ft;
					fp.wave_of_tab[ft] = (uint8_t)wv;
					++ft;
				}
			}
			fp.n_tabs = ft;
			/* Closed-form segments whose tables all sit in LDS in the wide form: more rows per pass where LDS holds them (round 4:
			 * a group's rows are contiguous in time since then, which took the 8-row build from 128 to 114 VGPRs) */
			if (use_fast && !seq_ok && wide_tabs_ && more_rows_ && FT == 8 && ft > 0 && ft == (uint32_t)__builtin_popcount(seg.wave_mask) &&
			    seg.max_steps >= 2) {
				for (uint32_t t : {12u, 10u}) /* (14 and 16 rows: hipcc's "requires even aligned vector registers" error on their spills) */
					if (t <= more_rows_ && ft * (size_t)FAST_TAB_BYTES_WIDE + 16 * area_of(t) + 1024 <= lds_limit_) {
						FT = t; area = area_of(FT); fp.rows = FT; fp.rows_multi = FT;
						break;
					}
			}
			/* the closed-form voices of a segment that may also have look-back voices: a launch of their own (FastParams.vlists) */
			const bool split_cf = use_fast && seq_ok && look_split;
			uint32_t rows_cf = 2;
			auto area_cf = [&](uint32_t t) { return (size_t)seg.n_fast * 64 * t * sizeof(float) + (size_t)fmax_steps * sizeof(unsigned long long); };
			if (split_cf) {
				for (uint32_t t : {8u, 6u, 5u, 4u})
					if (t <= fast_rows_ && ft * ftab_bytes + 16 * area_cf(t) + 1024 <= lds_limit_) { rows_cf = t; break; }
				/* (10 or 12 rows exist in the wide-table form only: where that fits -- the same test as wide_fits below) */
				if (rows_cf == 8 && wide_tabs_ && more_rows_ && ft > 0 && ft == (uint32_t)__builtin_popcount(seg.wave_mask) && seg.max_steps >= 2)
					for (uint32_t t : {12u, 10u})
						if (t <= more_rows_ && ft * (size_t)FAST_TAB_BYTES_WIDE + 16 * area_cf(t) + 1024 <= lds_limit_) { rows_cf = t; break; }
				if (!vlists_.ensure((size_t)2 * seg.n_voices, err)) return false;
				fp.vlists = vlists_.p; fp.split_cf = 1; fp.rows_cf = rows_cf;
				fp.look_words_real = look_words_real_ && fp.look ? 1u : 0u;
			}
			fp.cub_ok = use_fast && seg.maybe_cub && ft * ftab_bytes + 16 * area_cf(FAST_CUB_ROWS) + 1024 <= lds_limit_ &&
				(!look_split || ft * ftab_bytes + 16 * area_of(FAST_CUB_ROWS) + LOOK_LDS_BYTES + 1024 <= lds_limit_) ? 1u : 0u;
			/* the build for voices with chains and nothing to scan: as many rows per pass as fit beside the tables */
			fp.lean_on = chains && lean_enabled_ ? 1u : 0u;
			fp.rows_lean = 4;
			for (uint32_t t : {6u, 5u}) /* (decode_kernel lays the slots out for exactly the rows the build has: 4, 5 or 6) */
				if (t <= fast_rows_ && t <= lean_rows_ && ft * ftab_bytes + 16 * area_of(t) + 1024 <= lds_limit_) { fp.rows_lean = t; break; }
			if (ft * ftab_bytes + 16 * area_of(fp.rows_lean) + 1024 > lds_limit_) fp.lean_on = 0; /* (a tight LDS budget: the full build's rows) */
			TimedPair *ta = timing_on_ ? new_pair(3) : nullptr;
			if (ta) (void)hipEventRecord(ta->a, stream_);
			{
				/* a wave per voice: the voice's operator records and plan staged in LDS for lane 0's chain of dependent reads
				 * (k_analyze.h; round 6 -- a thread per voice on the records in HBM took 25-53 us per segment). Voices with more
				 * records or steps than 48 KiB hold are analysed on the records in HBM (SAU_AMD_ANALYZE_NO_LDS: all of them) */
				uint32_t lo = seg.max_ops, ls = seg.max_steps;
				size_t albytes = (size_t)lo * sizeof(DevOp) + (size_t)ls * sizeof(Step) + (size_t)lo * sizeof(uint32_t);
				static const bool no_lds = tune_env("SAU_AMD_ANALYZE_NO_LDS") != nullptr;
				if (albytes > 48 * 1024 || no_lds) { lo = 0; ls = 0; albytes = 0; }
				hipLaunchKernelGGL(analyze_kernel, dim3(seg.n_voices), dim3(64), albytes, stream_, fp, lo, ls);
			}
			if (ta) (void)hipEventRecord(ta->b, stream_);
			if (use_fast) {
				hipLaunchKernelGGL(decode_kernel, dim3(seg.n_voices), dim3(64), 0, stream_, fp);
				if (!wait_for_predecessor(err)) return false; /* (sauAmd_Batch_order_after: the rendering kernels from here on) */
				/* build 0: closed-form phases only; 1: every kind of running-sum voice; 2: single-pass voices and closed-form ones */
				const int main_build = !seq_ok ? 0 : look_split ? 2 : 1;
				/* (rows per pass 2, 4, 5, 6, 8; the full build never runs at more than 4: its 5- and 6-row slots stand in with 4's) */
				static const void *const fkernels[4][5] = {
					{(const void *)fast_kernel<2, 0>, (const void *)fast_kernel<4, 0>, (const void *)fast_kernel<5, 0>, (const void *)fast_kernel<6, 0>, (const void *)fast_kernel<8, 0>},
					{(const void *)fast_kernel<2, 1>, (const void *)fast_kernel<4, 1>, (const void *)fast_kernel<4, 1>, (const void *)fast_kernel<4, 1>, (const void *)fast_kernel<4, 1>},
					{(const void *)fast_kernel<2, 2>, (const void *)fast_kernel<4, 2>, (const void *)fast_kernel<5, 2>, (const void *)fast_kernel<6, 2>, (const void *)fast_kernel<8, 2>},
					/* 3: voices with feedback chains and nothing to scan (4, 5 or 6 rows per pass; 6 spills 170 VGPRs and is still the
					 * fastest on config 5 -- 52.3 against 53.1 ms per step at 5; the 8-row instantiation, 298 spilled, is gone since
					 * round 4, like the full build's 8-row one) */
					{(const void *)fast_kernel<4, 3>, (const void *)fast_kernel<4, 3>, (const void *)fast_kernel<5, 3>, (const void *)fast_kernel<6, 3>, (const void *)fast_kernel<6, 3>}};
				static size_t fconfigured[16][4][5];
				/* the closed-form build with wide table blocks in LDS (k_fast_types.h: FkTab), at 6 and 8 rows per pass: taken when
				 * every table the segment wants is in LDS and still fits in that form at the launch's rows */
				static const void *const fk_wide[4] = {(const void *)fast_kernel<6, 0, false, true>, (const void *)fast_kernel<8, 0, false, true>,
					(const void *)fast_kernel<10, 0, false, true>, (const void *)fast_kernel<12, 0, false, true>};
				static size_t wconfigured[16][4];
				const uint32_t n_want = (uint32_t)__builtin_popcount(seg.wave_mask);
				auto wide_fits = [&](uint32_t rows, size_t area16) {
					/* (voices of one step -- flat banks, BASELINE config 2 -- have no modulation to evaluate: fewer VALU
					 * instructions per sample to begin with, and the wider gather costs them 3-4 %) */
					return wide_tabs_ && (rows == 8 || rows == 6 || rows == 10 || rows == 12) && ft > 0 && ft == n_want && seg.max_steps >= 2 &&
						ft * (size_t)FAST_TAB_BYTES_WIDE + area16 + 1024 <= lds_limit_;
				};
				auto launch_build = [&](int build, uint32_t rows, uint32_t grid, const FastParams *prm = nullptr, size_t area16 = 0,
						bool wide = false) -> bool {
					if (build == 1 && rows > 4) rows = 4;
					if (build == 3) rows = rows < 5 ? 4 : rows > 6 ? 6 : rows;
					const int ri = rows == 8 ? 4 : rows == 6 ? 3 : rows == 5 ? 2 : rows == 4 ? 1 : 0;
					const size_t a16 = area16 ? area16 : 16 * area_of(rows);
					void *args[] = {(void *)(prm ? prm : &fp)};
					const bool wide_rows = rows == 8 || rows == 6 || rows == 10 || rows == 12;
					/* (a row count without an instantiation -- 10 or 12 rows outside the wide form, anything the
					 * tables above do not hold -- must not fall through to the 2-row kernel beside slots laid out for `rows`) */
					if (!(wide && build == 0 && wide_rows) && rows != 8 && rows != 6 && rows != 5 && rows != 4 && rows != 2) {
						err = "internal error: no fast_kernel build with " + std::to_string(rows) + " rows per pass in this form";
						return false;
					}
					if (build == 2 && rows == 8 && (wide || tail_live_)) { /* the look-back build with wide table blocks (round 5: where the lean buffer numbering leaves LDS for them), and / or the one that mixes few-voice streams (round 6: FastParams.tail_ok) */
						static size_t lwconfigured[16][3];
						const int li = wide ? (tail_live_ ? 2 : 0) : 1;
						const void *lk = wide ? (tail_live_ ? (const void *)fast_kernel<8, 2, false, true, true> : (const void *)fast_kernel<8, 2, false, true>)
						                      : (const void *)fast_kernel<8, 2, false, false, true>;
						const size_t lds = ft * (wide ? (size_t)FAST_TAB_BYTES_WIDE : ftab_bytes) + a16 + LOOK_LDS_BYTES;
						if (!raise_lds_attr(lk, lds, lwconfigured[dev_ & 15][li], err)) return false;
						HIP_OK(hipLaunchKernel(lk, dim3(grid), dim3(1024), args, lds, stream_));
						return true;
					}
					if (wide && build == 0 && wide_rows) {
						const int wi = rows == 12 ? 3 : rows == 10 ? 2 : rows == 8 ? 1 : 0;
						const size_t lds = ft * (size_t)FAST_TAB_BYTES_WIDE + a16;
						if (!raise_lds_attr(fk_wide[wi], lds, wconfigured[dev_ & 15][wi], err)) return false;
						HIP_OK(hipLaunchKernel(fk_wide[wi], dim3(grid), dim3(1024), args, lds, stream_));
						return true;
					}
					const size_t lds = ft * ftab_bytes + a16 + (build == 2 ? LOOK_LDS_BYTES : 0);
					if (!raise_lds_attr(fkernels[build][ri], lds, fconfigured[dev_ & 15][build][ri], err)) return false;
					HIP_OK(hipLaunchKernel(fkernels[build][ri], dim3(grid), dim3(1024), args, lds, stream_));
					return true;
				};
				/* (which form the closed-form launch of this segment takes -- repair_kernel runs in the same layout) */
				const bool wide_cf = main_build == 2 ? wide_fits(rows_cf, 16 * area_cf(rows_cf)) : main_build == 0 && wide_fits(FT, 16 * area);
				/* as many waves per voice as it has row groups (up to 64) when voices are
				 * few, one CU-filling grid at most */
				const uint32_t groups = (seg.len + (60 * FT) - 1) / (60 * FT);
				const unsigned long long want = (unsigned long long)seg.n_voices * (groups < 64 ? groups : 64);
				uint32_t fgrid = (uint32_t)((want + 15) / 16 > fk_grid_ ? fk_grid_ : (want + 15) / 16);
				if (fgrid > fk_grid_) fgrid = fk_grid_;
				if (fgrid < 1) fgrid = 1;
				TimedPair *tf = timing_on_ ? new_pair(2) : nullptr;
				if (tf) (void)hipEventRecord(tf->a, stream_);
				bool launched = true;
				auto launch_fast = [&](uint32_t mode, uint32_t grid = 0) {
					fp.mode = mode;
					if (fp.lean_on && fp.chain_rows && (mode == fp.sum_levels + 1 || mode == fp.sum_levels + 2) &&
					    !launch_build(3, fp.rows_lean, grid ? grid : fgrid))
						launched = false;
					if (main_build == 2) { /* what the single-pass build leaves out: returns at once when there is none */
						fp.only_multi = 1;
						if (!launch_build(1, FTM, grid ? grid : fgrid)) launched = false;
						fp.only_multi = 0;
					} else if (!launch_build(main_build, FT, grid ? grid : fgrid, nullptr, 0, wide_cf)) {
						launched = false;
					}
				};
				/* Tasks of a closed-form launch: G row groups each. The counter takes about one add per 12 ns chip-wide (r03:
				 * config 3 with tasks of 3 or 4 groups x 4 steps, one add per 8 ns, ran 1.3-1.6x slower), and a task of G groups
				 * x S steps keeps a wave for about G x S x 2.2 us, so G x S >= 48 leaves a factor of two; and dealing tasks
				 * out only pays when there are several per wave -- a short segment's few tasks go out in fixed strides, one
				 * contiguous run of groups per wave (1024 one-operator voices x 44100 frames: 0.15 ms in strides, 0.31 ms
				 * through the counter). */
				auto set_tasks = [&](FastParams &q, uint32_t n_groups, uint32_t grid) {
					const uint32_t waves = grid * 16;
					const uint32_t S = seg.max_steps ? seg.max_steps : 1;
					uint32_t G = dyn_groups_;
					if (G * S < dyn_floor_) G = (dyn_floor_ + S - 1) / S;
					const uint32_t k_dyn = (n_groups + G - 1) / G;
					if (dyn_enabled_ && (unsigned long long)seg.n_voices * k_dyn >= (unsigned long long)dyn_min_tasks_ * waves) {
						q.dyn_chunks = k_dyn ? k_dyn : 1;
						q.dyn_static = 0;
					} else {
						uint32_t k = waves / seg.n_voices;
						if (k > n_groups) k = n_groups;
						q.dyn_chunks = k ? k : 1;
						q.dyn_static = 1;
					}
				};
				FastParams cfp; /* the closed-form launch of a split segment (also what repair_kernel runs with) */
				tail_live_ = false;
				if (main_build == 2 && fp.look && FT == 8 && tailmix_enabled_ && max_write && !fp.cub_ok && seg.len < (1u << 30) &&
				    mix_few_enabled_ && max_rows <= 8 && seg.n_streams >= 8 && (seg.pcm_offset & 3u) == 0 && (pcm_row_ & 3u) == 0) {
					/* many streams of a few voices (a batch of small scripts): where a stream's last row is a look-back voice's and the
					 * others are closed-form voices', the look-back launch mixes the stream as it stores that row (k_fast_types.h:
					 * FastParams.tail_ok) and mix_few_kernel has nothing left to do (SAU_AMD_TAILMIX; off by default: see tailmix_enabled_) */
					if (!tail_ok_.ensure(seg.n_streams, err)) return false;
					fp.tail_ok = tail_ok_.p; fp.inmix_stream = S.mstreams.p;
					fp.tail_flags = (seg.stereo ? 1u : 0u) | (seg.swap_bytes ? 2u : 0u);
					fp.tail_pcm_offset = seg.pcm_offset;
					hipLaunchKernelGGL(tailmix_kernel, dim3((seg.n_streams + 63) / 64), dim3(64), 0, stream_, fp, seg.n_streams);
					tail_live_ = true;
				}
				if (main_build == 2) {
					/* two launches over analyze_kernel's lists: the closed-form voices, then the look-back voices */
					cfp = fp;
					cfp.n_fast = seg.n_fast; cfp.rows = rows_cf; cfp.mode = 0; cfp.only_multi = 0;
					const uint32_t groups_cf = (seg.len + (60 * rows_cf) - 1) / (60 * rows_cf);
					const unsigned long long want_cf = (unsigned long long)seg.n_voices * (groups_cf < 64 ? groups_cf : 64);
					const uint32_t grid_cf = (uint32_t)((want_cf + 15) / 16 > fk_grid_ ? fk_grid_ : (want_cf + 15) / 16);
					/* Both kinds of voice in ONE launch (duo_kernel, k_fast_voice.h; round 6) where both launches would run 8 rows per pass
					 * on narrow tables and the look-back voices' words in HBM exist (a voice then takes as many waves as there are). How
					 * many voices are of which kind only analyze_kernel knows: the kernel splits its waves by the lists' lengths */
					const bool wide_look_ = wide_tabs_ && !tune_env("SAU_AMD_NO_WIDE_LOOK") && FT == 8 && ft > 0 && ft == n_want &&
						ft * (size_t)FAST_TAB_BYTES_WIDE + 16 * area + LOOK_LDS_BYTES + 1024 <= lds_limit_; /* (the look-back launch would take wide tables: better apart) */
					const bool duo = duo_enabled_ && fp.look && fp.look_words_real && FT == 8 && rows_cf == 8 && !fp.cub_ok && !tail_live_ &&
						!wide_look_ && !wide_cf && ft * ftab_bytes + 16 * area + LOOK_LDS_BYTES + 1024 <= lds_limit_;
					if (getenv("SAU_AMD_DEBUG_DUO"))
						fprintf(stderr, "[sau-amd] duo %d: enabled %d look %d words_real %u FT %u rows_cf %u cub_ok %u tail %d may_scan %u of %u voices, lds %zu of %zu\n",
								(int)duo, (int)duo_enabled_, fp.look != nullptr, fp.look_words_real, FT, rows_cf, fp.cub_ok, (int)tail_live_, seg.n_may_scan, seg.n_voices,
								ft * ftab_bytes + 16 * area + LOOK_LDS_BYTES + 1024, lds_limit_);
					set_tasks(cfp, groups_cf, duo ? (fk_grid_ + 1) / 2 : grid_cf ? grid_cf : 1); /* (duo: half a workgroup's waves take closed-form tasks) */
					if (!duo && !launch_build(0, rows_cf, grid_cf ? grid_cf : 1, &cfp, 16 * area_cf(rows_cf), wide_cf)) launched = false;
					fp.mode = fp.sum_levels + 1; fp.only_multi = 0; fp.look_wpv = look_wpv_; fp.look_groups = groups;
					fp.look_wpv_flags = (tune_env("SAU_AMD_LOOK_NO_LDS") ? 1u : 0u) | (tune_env("SAU_AMD_LOOK_WITHHOLD") ? 2u : 0u) |
						(tune_env("SAU_AMD_LOOK_SPREAD") ? 4u : 0u);
					if (duo && tune_env("SAU_AMD_DUO_LW")) fp.look_wpv_flags |= ((uint32_t)atoi(tune_env("SAU_AMD_DUO_LW")) & 15u) << 8;
					if (duo && tune_env("SAU_AMD_DUO_PRIO")) fp.look_wpv_flags |= ((uint32_t)atoi(tune_env("SAU_AMD_DUO_PRIO")) & 7u) << 12; /* 1: none, 2: 1, 3: 2 (default), 4: 3 */
					if (duo) {
						static size_t duo_configured[16];
						const size_t dlds = ft * ftab_bytes + 16 * area + LOOK_LDS_BYTES;
						if (!raise_lds_attr((const void *)duo_kernel, dlds, duo_configured[dev_ & 15], err)) return false;
						auto go = [&]() -> bool {
							void *da[] = {(void *)&fp, (void *)&cfp};
							return hipLaunchKernel((const void *)duo_kernel, dim3(fk_grid_), dim3(1024), da, dlds, stream_) == hipSuccess;
						};
						/* (its look-back waves wait for sums across workgroups: in turn with other such launches of the process) */
						if (!SpreadLaunchOrder::get().ordered(dev_, stream_, go)) launched = false;
					} else
					if (fp.look) {
						const unsigned long long waves = (unsigned long long)seg.n_voices * (fp.look_words_real ? (groups < 64 ? groups : 64) : look_wpv_);
						const uint32_t grid2 = waves > (unsigned long long)fk_grid_ * 16 ? fk_grid_ : (uint32_t)((waves + 15) / 16);
						/* (wide table blocks for the look-back voices too, where they fit beside the buffers at 8 rows: the carrier-FM bank) */
						const bool wide_look = wide_tabs_ && !tune_env("SAU_AMD_NO_WIDE_LOOK") && FT == 8 && ft > 0 && ft == n_want &&
							ft * (size_t)FAST_TAB_BYTES_WIDE + 16 * area + LOOK_LDS_BYTES + 1024 <= lds_limit_;
						if (look_inside_) { /* (every voice within one workgroup: no waits across workgroups) */
							if (!launch_build(2, FT, grid2 ? grid2 : 1, nullptr, 0, wide_look)) launched = false;
						} else if (!SpreadLaunchOrder::get().ordered(dev_, stream_, [&]() { return launch_build(2, FT, grid2 ? grid2 : 1, nullptr, 0, wide_look); })) {
							launched = false;
						}
						if (fp.cub_ok) { /* look-back voices with the loop tails of `cub` R segments: the build with that code, same lists */
							static size_t cub2_configured[16];
							const void *ck2 = (const void *)fast_kernel<(int)FAST_CUB_ROWS, 2, true>;
							const size_t lds2 = ft * ftab_bytes + 16 * area_of(FAST_CUB_ROWS) + LOOK_LDS_BYTES;
							if (!raise_lds_attr(ck2, lds2, cub2_configured[dev_ & 15], err)) return false;
							auto go = [&]() -> bool {
								void *a2[] = {(void *)&fp};
								return hipLaunchKernel(ck2, dim3(grid2 ? grid2 : 1), dim3(1024), a2, lds2, stream_) == hipSuccess;
							};
							if (look_inside_ ? !go() : !SpreadLaunchOrder::get().ordered(dev_, stream_, go)) launched = false;
						}
					}
				}
				if (fp.scan) {
					/* some voice may have running-sum phases: sums per row group, their prefixes, final pass */
					/* (with look-back only voices that have, or may get, feedback chains still take sum passes) */
					if (fp.chain_rows && fp.chain_early_ok) {
						/* chains that sums or other chains' inputs depend on, fed from their own lines: whole segment, before
						 * anything else (FastInfo.early); returns at once when analyze_kernel found none */
						const size_t clds = chain_alone_lds((size_t)fp.n_ctabs * CHAIN_TAB_BYTES + CHAIN_IO_BYTES, (seg.n_chain_slots + 63) / 64);
						if (!raise_lds_attr((const void *)chain_kernel, clds, chain_lds_configured_[dev_ & 15], err)) return false;
						fp.chain_early = 1; fp.range_mode = 0;
						hipLaunchKernelGGL(chain_kernel, dim3((seg.n_chain_slots + 63) / 64), dim3(192), clds, stream_, fp);
						{ /* R feedback: a chain wave and a feeder wave per 64 chains (k_chain.h) */
							static size_t rc_configured[16];
							const size_t rlds_ = chain_alone_lds(RCHAIN_LDS_BYTES, (seg.n_chain_slots + 63) / 64);
							if (!raise_lds_attr((const void *)rchain_kernel, rlds_, rc_configured[dev_ & 15], err)) return false;
							hipLaunchKernelGGL(rchain_kernel, dim3((seg.n_chain_slots + 63) / 64), dim3(RCHAIN_THREADS), rlds_, stream_, fp);
						}
						fp.chain_early = 0;
					}
					for (uint32_t pass = 1; pass <= fp.sum_levels && (!fp.look || seg.n_chain_rows); ++pass) {
						launch_fast(pass);
						hipLaunchKernelGGL(scan_kernel, dim3(seg.n_voices), dim3(64), 0, stream_, fp);
					}
					if (fp.chain_rows) {
						/* The chains' inputs, the chains themselves (lanes = voices), the final pass -- pipelined over
						 * chunks of the segment: chain_kernel occupies one CU per 64 chains for frames x chain latency,
						 * so it runs on a stream of its own while, on the other CUs, the chain-input pass prepares the
						 * chunks after it and the final pass finishes the chunks before it. */
						const uint32_t cgrid = (seg.n_chain_slots + 63) / 64;
						const size_t clds = chain_alone_lds((size_t)fp.n_ctabs * CHAIN_TAB_BYTES + CHAIN_IO_BYTES, cgrid);
						if (!raise_lds_attr((const void *)chain_kernel, clds, chain_lds_configured_[dev_ & 15], err)) return false;
						/* chunks of about chain_chunk_frames_ frames (what the first chunk's inputs and the last chunk's final
						 * pass take is not overlapped with the chains), 32 at most */
						uint32_t n_chunks = chain_chunks_ ? chain_chunks_ : (seg.len + chain_chunk_frames_ - 1) / chain_chunk_frames_;
						if (n_chunks > 32) n_chunks = 32;
						while (n_chunks > 1 && seg.len / n_chunks < 4096) --n_chunks;
						if (!chain_stream_ && n_chunks > 1) {
							chain_stream_ = StreamPool::get().take(dev_, 1);
							if (!chain_stream_) {
								/* High priority, for the sake of the hardware queue it gets: the runtime maps streams onto a
								 * handful of queues (four per priority), and a second generator's two streams of ordinary priority
								 * landed on ONE -- its chains and its passes then ran one after the other, 76 ms for a config-5
								 * step instead of 45 (profiles/r05_c5_pipelined_trace.txt). Queues of another priority are others:
								 * a generator's chain stream never shares its main stream's. (The chain wave is also what the
								 * step's time hangs on.) */
								int lo = 0, hi = 0;
								(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
								HIP_OK(hipStreamCreateWithPriority(&chain_stream_, hipStreamNonBlocking, hi));
							}
						}
						/* chunk boundaries: multiples of the chain kernel's batch. What follows the last chunk's chains is not
						 * overlapped with anything -- its final pass, finalize_kernel, its share of the mixer (which follows the
						 * other chunks since round 5) -- so the last chunk is a short one (4096 frames: 0.4 ms of chains, 0.15 ms of
						 * final pass behind them instead of 0.5-0.7) */
						uint32_t cb[34];
						{
							const uint32_t tail = (!chain_chunks_ && short_last_chunk_ && n_chunks >= 4 && n_chunks < 32 && seg.len > 8 * 4096u) ? 4096u : 0u;
							const uint32_t body = (seg.len - tail) & ~255u;
							const uint32_t clen0 = (((tail ? body : seg.len) + n_chunks - 1) / n_chunks + 255) & ~255u;
							for (uint32_t c = 0; c <= n_chunks; ++c) cb[c] = c * clen0 < (tail ? body : seg.len) ? c * clen0 : (tail ? body : c * clen0);
							if (tail) { cb[n_chunks] = body; ++n_chunks; cb[n_chunks] = seg.len; }
							/* (with the rounded-up chunk length the last body chunks can come out empty -- seg.len 81930 in 20
							 * chunks: cb[19] == cb[20] -- and each would still cost a chain launch, two pass launches and two events) */
							uint32_t m = 0;
							for (uint32_t c = 0; c < n_chunks; ++c)
								if (cb[c + 1] > cb[c]) { cb[m] = cb[c]; ++m; cb[m] = cb[c + 1]; }
							if (m) n_chunks = m;
						}
						while (chain_ev_.size() < 2 * (size_t)n_chunks) {
							hipEvent_t e;
							HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
							chain_ev_.push_back(e);
						}
						/* the time-parallel passes leave the chains' CUs alone while both run */
						const uint32_t pgrid = n_chunks > 1 && fgrid + cgrid > fk_grid_ ? (fk_grid_ > cgrid + 32 ? fk_grid_ - cgrid : 32) : fgrid;
						TimedPair *tc = timing_on_ ? new_pair(0) : nullptr; /* counted with the block loop it replaces */
						if (n_chunks == 1) {
							fp.range_mode = 0;
							launch_fast(fp.sum_levels + 2);
							if (tc) (void)hipEventRecord(tc->a, stream_);
							hipLaunchKernelGGL(chain_kernel, dim3(cgrid), dim3(192), clds, stream_, fp);
							if (tc) (void)hipEventRecord(tc->b, stream_);
							launch_fast(fp.sum_levels + 1);
						} else {
							for (uint32_t c = 0; c < n_chunks; ++c) { /* inputs of chunk c, then its chains on the other stream */
								fp.range_mode = 1; fp.f_lo = cb[c]; fp.f_hi = c + 1 == n_chunks ? 0xffffffffu : cb[c + 1];
								fp.range_last = c + 1 == n_chunks;
								launch_fast(fp.sum_levels + 2, pgrid);
								HIP_OK(hipEventRecord(chain_ev_[2 * c], stream_));
								HIP_OK(hipStreamWaitEvent(chain_stream_, chain_ev_[2 * c], 0));
								FastParams cp = fp;
								cp.range_mode = 1; cp.f_lo = cb[c]; cp.f_hi = c + 1 == n_chunks ? 0xffffffffu : cb[c + 1];
								if (tc && c == 0) (void)hipEventRecord(tc->a, chain_stream_);
								hipLaunchKernelGGL(chain_kernel, dim3(cgrid), dim3(192), clds, chain_stream_, cp);
								if (tc && c + 1 == n_chunks) (void)hipEventRecord(tc->b, chain_stream_);
								HIP_OK(hipEventRecord(chain_ev_[2 * c + 1], chain_stream_));
							}
							/* the mixer follows the final passes chunk by chunk (k_finish.h: premix_kernel): after the pass over the
							 * row groups that end in chunk c, every frame below the chunk's end less one group's span is written
							 * (a group has at most 64 x 6 frames in the builds that run here) */
							/* (not where a launch BEHIND the chunks still writes voice rows: the closed-form build with the loop tails of
							 * `cub` R segments, below -- found by round 5's drop-in sweep, 2 programs of 3000; rows that repair_kernel
							 * touches afterwards make the last launch mix everything again: fast_voice notes it in work_count[1]) */
							const bool early_mix = early_mix_enabled_ && max_write && !fp.cub_ok && !(mix_few_enabled_ && max_rows <= 8 && seg.n_streams >= 8);
							uint32_t mixed_blocks = 0;
							if (early_mix) hipLaunchKernelGGL(premix_kernel, dim3((seg.n_voices + 63) / 64), dim3(64), 0, stream_, fp);
							for (uint32_t c = 0; c < n_chunks; ++c) { /* the final pass follows the chains chunk by chunk */
								HIP_OK(hipStreamWaitEvent(stream_, chain_ev_[2 * c + 1], 0));
								fp.range_mode = 2; fp.f_lo = cb[c]; fp.f_hi = c + 1 == n_chunks ? 0xffffffffu : cb[c + 1];
								fp.range_last = c + 1 == n_chunks;
								launch_fast(fp.sum_levels + 1, c + 1 == n_chunks ? fgrid : pgrid);
								if (early_mix && c + 1 < n_chunks && cb[c + 1] > 512) {
									const uint32_t hi = (cb[c + 1] - 512) / 256;
									if (hi > mixed_blocks) {
										MixParams mp = mix_params(seg, S);
										mp.blk_lo = mixed_blocks; mp.blk_hi = hi;
										TimedPair *tm = timing_on_ ? new_pair(1) : nullptr;
										if (tm) (void)hipEventRecord(tm->a, stream_);
										hipLaunchKernelGGL(mix_kernel, dim3(hi - mixed_blocks, seg.n_streams), dim3(256), 0, stream_, mp);
										if (tm) (void)hipEventRecord(tm->b, stream_);
										mixed_blocks = hi;
									}
								}
							}
							early_mixed_blocks_ = mixed_blocks;
							fp.range_mode = 0; fp.f_lo = 0; fp.f_hi = 0; fp.range_last = 0;
						}
					} else {
						launch_fast(fp.sum_levels + 1);
					}
				} else {
					/* closed-form voices only: tasks of about eight row groups, dealt out by a counter */
					if (main_build == 0) set_tasks(fp, groups, fgrid);
					/* tasks dealt out by the counter: one queue per XCD, chunk-major (k_fast_types.h) -- 1.80 -> 1.76 ms per config-3
					 * launch against the one counter in voice order */
					if (fp.inmix && main_build == 0 && !fp.dyn_static && fp.dyn_chunks >= 16 && !fp.cub_ok) {
						fp.inmix_flags = 64u; /* (analyze_kernel has cleared the queues) */
						/* ... whose last eight chunks are short ones (k_fast_types.h: INMIX_NCH1): a quarter of a task's row groups, where the
						 * segment has at least four regular chunks per XCD besides */
						if (inmix_taper_) {
							const uint32_t S_ = seg.max_steps ? seg.max_steps : 1;
							uint32_t G = dyn_groups_;
							if (G * S_ < dyn_floor_) G = (dyn_floor_ + S_ - 1) / S_;
							const uint32_t small = G / 4 ? G / 4 : 1;
							if (groups >= 32 * G + INMIX_NSMALL * small) {
								fp.dyn_small = small;
								fp.dyn_chunks = (groups - INMIX_NSMALL * small + G - 1) / G + INMIX_NSMALL;
							}
						}
						{ /* the voice count as a divisor (k_fast_voice.h: udiv_magic) */
							const uint32_t d = seg.n_voices;
							uint32_t sh = 0;
							while ((1ull << sh) < d) ++sh;
							fp.inmix_div_s = sh;
							fp.inmix_div_m = (uint32_t)(((1ull << (32 + sh)) / d) + 1ull - (1ull << 32));
						}
						/* ... and a bank of voices mixed into one stream: the launch mixes its own rows, chunk by chunk behind the rendering;
						 * premix_kernel has the last word. Config 3: 2.056 -> 1.995 ms per step (the launch 1.74 -> 1.87 ms, the mixer
						 * 0.26 -> 0.07 ms) */
						if (inmix_enabled_ && seg.n_streams == 1 && max_write && seg.n_voices >= inmix_min_voices_ && seg.max_steps >= inmix_min_steps_ &&
						    row_stride_ < (1u << 24)) {
							fp.inmix_stream = S.mstreams.p;
							fp.inmix_flags = 64u | 32u | (seg.stereo ? 1u : 0u) | (seg.swap_bytes ? 2u : 0u) | (tune_env("SAU_AMD_INMIX_DRY") ? 4u : 0u) | ((inmix_at_ & 15u) << 8);
							fp.inmix_pcm_offset = seg.pcm_offset;
							hipLaunchKernelGGL(premix_kernel, dim3((seg.n_voices + 63) / 64), dim3(64), 0, stream_, fp);
							inmix_live_ = true;
						}
					}
					if (inner_enabled_ && main_build == 0 && wide_cf && FT == 12 && fp.dyn_chunks >= 3) {
						/* BASELINE config 3's build in two launches (k_fast_voice.h: INNER): every voice's first and last row group by the plain
						 * build, a wave per group; then the groups between by the build that holds only the form without in-segment masks */
						static size_t inner_configured[16];
						const void *ik = (const void *)fast_kernel<12, 0, false, true, false, true>;
						const size_t lds = ft * (size_t)FAST_TAB_BYTES_WIDE + 16 * area;
						FastParams ep = fp;
						ep.mode = 0; ep.edge_only = 1; ep.dyn_static = 1; ep.dyn_chunks = 2; ep.dyn_small = 0; ep.inmix_flags = 0; /* (two tasks a voice: its first group, its last) */
						const uint32_t egrid = (2 * seg.n_voices + 15) / 16 < fk_grid_ ? (2 * seg.n_voices + 15) / 16 : fk_grid_;
						if (!launch_build(0, FT, egrid, &ep, 0, true)) launched = false;
						fp.mode = 0;
						void *iargs[] = {(void *)&fp};
						if (!raise_lds_attr(ik, lds, inner_configured[dev_ & 15], err)) return false;
						HIP_OK(hipLaunchKernel(ik, dim3(fgrid), dim3(1024), iargs, lds, stream_));
					} else {
						launch_fast(0);
					}
					fp.dyn_chunks = 0; fp.inmix_flags = 0; fp.dyn_small = 0;
				}
				{ /* closed-form voices with the loop tails of `cub` R segments (FastInfo.cub): the build with that code,
				   * FAST_CUB_ROWS rows per pass, fixed strides over the same voices as the closed-form launch; returns at
				   * once when analyze_kernel found none */
					static size_t cub_configured[16];
					FastParams q = main_build == 2 ? cfp : fp;
					q.n_fast = seg.n_fast; q.rows = FAST_CUB_ROWS; q.mode = 0; q.only_multi = 0;
					const size_t qlds = ft * ftab_bytes + 16 * area_cf(FAST_CUB_ROWS);
					if (fp.cub_ok) {
						const uint32_t g4 = (seg.len + (60 * FAST_CUB_ROWS) - 1) / (60 * FAST_CUB_ROWS);
						const unsigned long long want4 = (unsigned long long)seg.n_voices * (g4 < 64 ? g4 : 64);
						const uint32_t grid4 = (uint32_t)((want4 + 15) / 16 > fk_grid_ ? fk_grid_ : (want4 + 15) / 16);
						uint32_t k = (grid4 ? grid4 : 1) * 16 / seg.n_voices;
						if (k > g4) k = g4;
						q.dyn_chunks = k ? k : 1; q.dyn_static = 1;
						const void *ck = (const void *)fast_kernel<(int)FAST_CUB_ROWS, 0, true>;
						if (!raise_lds_attr(ck, qlds, cub_configured[dev_ & 15], err)) return false;
						void *qargs[] = {(void *)&q};
						HIP_OK(hipLaunchKernel(ck, dim3(grid4 ? grid4 : 1), dim3(1024), qargs, qlds, stream_));
					}
				}
				if (!launched) return false;
				{ /* row groups noted for a second evaluation: returns at once when there are none */
					/* (only closed-form voices note any: with split launches they run at rows_cf rows per pass, in cfp's layout) */
					const uint32_t RT = main_build == 2 ? rows_cf : FT;
					const size_t rtab = wide_cf ? (size_t)FAST_TAB_BYTES_WIDE : ftab_bytes;
					const size_t rlds = main_build == 2 ? ft * rtab + 16 * area_cf(rows_cf) : ft * rtab + 16 * area;
					FastParams rpar = main_build == 2 ? cfp : fp;
					rpar.mode = 0;
					const void *rk = wide_cf && RT == 12 ? (const void *)repair_kernel<12, true> : wide_cf && RT == 10 ? (const void *)repair_kernel<10, true>
					               : wide_cf && RT == 8 ? (const void *)repair_kernel<8, true> : wide_cf && RT == 6 ? (const void *)repair_kernel<6, true>
					               : RT == 8 ? (const void *)repair_kernel<8> : RT == 6 ? (const void *)repair_kernel<6>
					               : RT == 5 ? (const void *)repair_kernel<5> : RT == 4 ? (const void *)repair_kernel<4>
					               : (const void *)repair_kernel<2>;
					static size_t rconfigured[16][9];
					if (!raise_lds_attr(rk, rlds, rconfigured[dev_ & 15][wide_cf && RT == 12 ? 8 : wide_cf && RT == 10 ? 7 : wide_cf && RT == 8 ? 6 : wide_cf && RT == 6 ? 5 : RT == 8 ? 4 : RT == 6 ? 3 : RT == 5 ? 2 : RT == 4 ? 1 : 0], err)) return false;
					const uint32_t rgrid = (seg.n_voices + 15) / 16 < 64 ? (seg.n_voices + 15) / 16 : 64;
					fp.mode = 0;
					void *rargs[] = {(void *)&rpar};
					HIP_OK(hipLaunchKernel(rk, dim3(rgrid), dim3(1024), rargs, rlds, stream_));
				}
				if (tf) (void)hipEventRecord(tf->b, stream_);
			}
			TimedPair *tz = timing_on_ ? new_pair(3) : nullptr;
			if (tz) (void)hipEventRecord(tz->a, stream_);
			hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)(((size_t)seg.n_voices * fp.max_ops * 8 + 63) / 64)), dim3(64), 0,
					stream_, fp);
			if (tz) (void)hipEventRecord(tz->b, stream_);
			HIP_OK(hipGetLastError());
			if (debug_ && use_fast) {
				(void)hipStreamSynchronize(stream_);
				FastInfo fi0;
				(void)hipMemcpy(&fi0, finfo_.p, sizeof fi0, hipMemcpyDeviceToHost);
				fprintf(stderr, "[sau-amd] fast: voice 0 total %u H %u bail %u steps %u seq %u; n_fast %u rows %u\n",
						fi0.total, fi0.H, fi0.bail, fi0.n_fsteps, fi0.seq, n_fast, FT);
				std::vector<FastStep> fs(fi0.n_fsteps < 64 ? fi0.n_fsteps : 64);
				if (!fs.empty()) (void)hipMemcpy(fs.data(), fsteps_.p, fs.size() * sizeof(FastStep), hipMemcpyDeviceToHost);
				for (size_t i = 0; i < fs.size(); ++i)
					fprintf(stderr, "[sau-amd]   step %zu kind %u flags %#x which %u dep %u out %d pm %d fpm %d amp %d aux %d type %#x inc %u ac %g fc %g ramp %u\n",
							i, fs[i].kind & 0xff, (fs[i].kind >> 8) & 0xff, (fs[i].kind >> 16) & 0xff, fs[i].kind >> 24,
							(int)fs[i].out_off, (int)fs[i].pm_off, (int)fs[i].fpm_off, (int)fs[i].amp_off, (int)fs[i].aux_off,
							fs[i].type, fs[i].inc, fs[i].ac, fs[i].fc, fs[i].ramp);
			}
			rp.fast_done = fdone_.p; rp.worklist = worklist_.p; rp.work_count = work_count_.p;
			/* Block-loop grid: persistent over the device-built work list. When the
			 * host knows of nothing that needs it (no sweeps, FM, feedback or expiring
			 * operators), a token grid still serves the rare dphase == 0 bail-out. */
			if (V > 1) {
				const uint32_t full = (seg.n_voices + V - 1) / V;
				block_grid_ = (seg.maybe_block || !use_fast) ? (full < 512 ? full : 512) : 2;
			} else {
				block_grid_ = (seg.maybe_block || !use_fast) ? (seg.n_voices < 1024 ? seg.n_voices : 1024) : 16;
			}
		}
		if (HB) {
			const size_t stride = ((size_t)seg.n_slots * 64 * sizeof(float) + (size_t)seg.max_ops * sizeof(DevOp) +
				(size_t)seg.max_steps * sizeof(Step) + 255) / 256 * 256 / sizeof(float);
			if (stride > 0xffffffffu) { err = "voice too large (block buffers + operator records beyond 16 GiB)"; return false; }
			if (!big_slots_.ensure((size_t)block_grid_ * stride, err)) return false;
			rp.big_slots = big_slots_.p; rp.big_stride = (uint32_t)stride;
		}
		if (!wait_for_predecessor(err)) return false; /* (a segment without the time-parallel path) */
		TimedPair *tp = timing_on_ ? new_pair(0) : nullptr;
		if (tp) (void)hipEventRecord(tp->a, stream_);
		bool ok = V == 8 ? (T == 4 ? launch_render<1, 4, 8>(rp, block_grid_, lds, err)
		                  : T == 3 ? launch_render<1, 3, 8>(rp, block_grid_, lds, err)
		                  : T == 2 ? launch_render<1, 2, 8>(rp, block_grid_, lds, err)
		                           : launch_render<1, 1, 8>(rp, block_grid_, lds, err))
		        : V > 1 ? (T == 4 ? launch_render<1, 4, 16>(rp, block_grid_, lds, err)
		                 : T == 3 ? launch_render<1, 3, 16>(rp, block_grid_, lds, err)
		                 : T == 2 ? launch_render<1, 2, 16>(rp, block_grid_, lds, err)
		                          : launch_render<1, 1, 16>(rp, block_grid_, lds, err))
		        : HB ? launch_render<1, 1, 1, true>(rp, block_grid_, lds, err)
		        : (W == 1) ? launch_render<1, 1, 1>(rp, block_grid_, lds, err)
		        : geo_ ? launch_render<4, 4, 1>(rp, block_grid_, lds, err)
		               : launch_render<8, 2, 1>(rp, block_grid_, lds, err);
		if (!ok) return false;
		if (tp) (void)hipEventRecord(tp->b, stream_);
		if (debug_) debug_dump("after render", seg);
		if (max_write) {
			MixParams mp = mix_params(seg, S);
			/* (frames an early launch has mixed are skipped when the device says those results stand: k_finish.h) */
			mp.early_blocks = early_mixed_blocks_;
			early_mixed_blocks_ = 0;
			mp.inmix = inmix_live_ ? inmix_ctl_.p : nullptr; /* (the closed-form launch has mixed tiles itself: mix_kernel takes what is left) */
			mp.tail_ok = tail_live_ ? tail_ok_.p : nullptr; /* (the look-back launch has mixed some streams itself: mix_few_kernel leaves them) */
			tail_live_ = false;
			inmix_live_ = false;
			/* (on the generator's one stream, behind the segment's kernels. Round 3 built the mixer on a stream of its own beside the
			 * next segment's kernels -- ordinary grid or persistent on a few CUs, everything it reads and writes double-buffered --
			 * and measured it no faster in any form, profiles/r03_headline_ab.json; that code is gone since round 4.) */
			hipStream_t ms = stream_;
			TimedPair *tm = timing_on_ ? new_pair(1) : nullptr;
			if (tm) (void)hipEventRecord(tm->a, ms);
			if (mix_few_enabled_ && max_rows <= 8 && seg.n_streams >= 8 && (seg.pcm_offset & 3u) == 0 && (pcm_row_ & 3u) == 0)
				/* many streams of a few voices each: four frames per thread, no tile staging (k_finish.h) */
				hipLaunchKernelGGL(mix_few_kernel, dim3((max_write + 1023) / 1024, seg.n_streams), dim3(256), 0, ms, mp);
			else if (mp.inmix && !mp.early_blocks && mix64_enabled_)
				hipLaunchKernelGGL(mix_kernel64, dim3((max_write + 63) / 64, seg.n_streams), dim3(64), 0, ms, mp);
			else
				hipLaunchKernelGGL(mix_kernel, dim3((max_write + 255) / 256, seg.n_streams), dim3(256), 0, ms, mp);
			HIP_OK(hipGetLastError());
			if (tm) (void)hipEventRecord(tm->b, ms);
			if (mp.inmix && inmix_report_) { /* (tests: what the launch mixed itself) */
				std::vector<uint32_t> h(INMIX_WORDS);
				uint32_t g[2] = {0, 0};
				(void)hipStreamSynchronize(ms);
				(void)hipMemcpy(h.data(), mp.inmix, INMIX_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost);
				(void)hipMemcpy(g, work_count_.p, sizeof g, hipMemcpyDeviceToHost);
				const uint32_t nch = h[INMIX_NCH] < INMIX_MAX_CHUNKS ? h[INMIX_NCH] : INMIX_MAX_CHUNKS, tpc = h[INMIX_TPC];
				uint32_t tiles = 0, whole = 0, all = 0;
				for (uint32_t k = 0; k < nch; ++k) {
					/* (the chunk's tiles: a regular chunk's, or a short one's at the end -- k_fast_types.h: INMIX_NCH1) */
					uint32_t lo_f = k * h[INMIX_CF], hi_f = lo_f + h[INMIX_CF];
					if (k >= h[INMIX_NCH1]) { lo_f = h[INMIX_BASE] + (k - h[INMIX_NCH1]) * h[INMIX_CFS]; hi_f = lo_f + h[INMIX_CFS]; }
					else if (hi_f > h[INMIX_BASE]) hi_f = h[INMIX_BASE];
					if (hi_f > max_write) hi_f = max_write;
					const uint32_t tk = hi_f > lo_f ? (hi_f - lo_f + INMIX_TILE - 1) / INMIX_TILE : 0u;
					uint32_t c = 0;
					for (uint32_t j = 0; j < tk && j < tpc && j < INMIX_MAX_TPC; ++j) c += (h[INMIX_CHUNK + INMIX_LINE * k + INMIX_BITS + (j >> 5)] >> (j & 31u)) & 1u;
					tiles += c; whole += c == tk ? 1u : 0u; all += tk;
				}
				fprintf(stderr, "[sau-amd] inmix: voices %u frames %u chunks %u x %u frames, tiles %u of %u, chunks mixed whole %u, guard %u %u\n",
						seg.n_voices, max_write, nch, h[INMIX_CF], tiles, all, whole, g[0], g[1]);
			}
		}
		return true;
	}

	template <typename SetT> MixParams mix_params(const SegmentDesc &seg, const SetT &S) {
		MixParams mp;
		mp.streams = S.mstreams.p; mp.vout = S.vout.p; mp.pan = S.pan.p; mp.vinfo = S.vinfo.p;
		mp.row_stride = row_stride_; mp.pcm_offset = seg.pcm_offset;
		mp.stereo = seg.stereo ? 1 : 0;
		mp.swap_bytes = seg.swap_bytes ? 1 : 0;
		mp.blk_lo = 0; mp.blk_hi = 0; mp.early_blocks = 0; mp.guard = work_count_.p; mp.inmix = nullptr; mp.tail_ok = nullptr;
		return mp;
	}

	bool fetch_pcm(uint32_t stream, int16_t *dst, uint32_t frames, bool stereo, std::string &err) override {
		use_device();
		const size_t n = (size_t)frames * (stereo ? 2 : 1);
		/* the caller's memory is pageable: a device copy straight into it costs milliseconds of
		 * pinning per call, so the PCM goes through a page-locked block (unless dst is one) */
		const bool pinned = host_blocks_.count(dst) != 0;
		if (!pinned && !h_pcm_.ensure(n, err)) return false;
		HIP_OK(hipMemcpyAsync(pinned ? dst : h_pcm_.p, pcm_.p + pcm_row_ * stream, n * sizeof(int16_t),
				hipMemcpyDeviceToHost, stream_));
		HIP_OK(hipStreamSynchronize(stream_));
		if (!pinned) memcpy(dst, h_pcm_.p, n * sizeof(int16_t));
		return true;
	}

	/* output stage: copies into page-locked memory queue behind the mixer and
	 * ahead of the next run's kernels on the one stream */
	bool fetch_pcm_async(uint32_t stream, int16_t *dst, uint32_t frames, bool stereo, int slot,
			std::string &err) override {
		slot &= 3;
		use_device();
		if (!fetch_ev_[slot]) HIP_OK(hipEventCreateWithFlags(&fetch_ev_[slot], hipEventDisableTiming));
		hipStream_t cs = stream_;
		HIP_OK(hipMemcpyAsync(dst, pcm_.p + pcm_row_ * stream,
				(size_t)frames * (stereo ? 2 : 1) * sizeof(int16_t), hipMemcpyDeviceToHost, cs));
		HIP_OK(hipEventRecord(fetch_ev_[slot], cs));
		return true;
	}
	bool wait_fetch(int slot, std::string &err) override {
		slot &= 3;
		if (fetch_ev_[slot]) HIP_OK(hipEventSynchronize(fetch_ev_[slot]));
		return true;
	}
	void *alloc_host(size_t bytes) override {
		use_device();
		std::string err;
		void *p = pool_alloc(true, bytes, err);
		if (p) host_blocks_[p] = bytes;
		return p;
	}
	void free_host(void *p) override {
		use_device();
		auto it = host_blocks_.find(p);
		if (it == host_blocks_.end()) return;
		(void)hipStreamSynchronize(stream_);
		pool_free(true, p, it->second);
		host_blocks_.erase(it);
	}

	const int16_t *device_pcm(uint32_t stream) override { return pcm_.p ? pcm_.p + pcm_row_ * stream : nullptr; }

	bool sync(std::string &err) override {
		use_device();
		HIP_OK(hipStreamSynchronize(stream_));
		arena_used_ = 0; /* every staged copy has left the arena */
		return true;
	}
	/* the operator records are all the state a generator has on the device (everything else is per segment): a copy of them
	 * on the generator's stream, behind the kernels of the run before and ahead of those of the run after */
	bool save_state(int slot, std::string &err) override {
		slot &= 3;
		use_device();
		if (!ops_snap_[slot].ensure(ops_.cap, err)) return false;
		HIP_OK(hipMemcpyAsync(ops_snap_[slot].p, ops_.p, ops_.cap * sizeof(DevOp), hipMemcpyDeviceToDevice, stream_));
		return true;
	}
	bool load_state(int slot, std::string &err) override {
		slot &= 3;
		use_device();
		if (!ops_snap_[slot].p || ops_snap_[slot].cap < ops_.cap) { err = "no operator state was saved under this slot"; return false; }
		HIP_OK(hipMemcpyAsync(ops_.p, ops_snap_[slot].p, ops_.cap * sizeof(DevOp), hipMemcpyDeviceToDevice, stream_));
		return true;
	}

	void timing(double *render_ms, double *mix_ms, uint64_t *launches, bool reset) override {
		if (!timing_on_) { timing_on_ = true; }
		(void)hipStreamSynchronize(stream_);
		drain_pairs();
		if (render_ms) *render_ms = acc_ms_[0] + acc_ms_[2];
		if (mix_ms) *mix_ms = acc_ms_[1];
		if (launches) *launches = acc_launches_;
		if (reset) { acc_ms_[0] = acc_ms_[1] = acc_ms_[2] = acc_ms_[3] = 0; acc_launches_ = 0; }
	}

	size_t chain_rows_budget() override { return sauengine::chain_rows_budget(free_hint_, chain_rows_failures_); }
	void *stream_handle() override { return (void *)stream_; }
	bool order_after(HipBackend *before, std::string &err) override {
		HipBackendImpl *o = static_cast<HipBackendImpl *>(before);
		if (o == this) return true;
		if (o->dev_ != dev_) { err = "sauAmd_Batch_order_after: the two batches are on different devices"; return false; }
		use_device();
		if (!after_ev_) HIP_OK(hipEventCreateWithFlags(&after_ev_, hipEventDisableTiming));
		/* (everything of `before` ends on its main stream: its chain stream's work is joined there by the final passes) */
		HIP_OK(hipEventRecord(after_ev_, o->stream_));
		after_pending_ = true;
		return true;
	}
	/* (render(): ahead of the first kernel that does real work -- after analyze_kernel and decode_kernel) */
	bool wait_for_predecessor(std::string &err) {
		if (!after_pending_) return true;
		after_pending_ = false;
		HIP_OK(hipStreamWaitEvent(stream_, after_ev_, 0));
		return true;
	}
	void set_timing(int level) override { timing_on_ = level > 0; timing_level_ = level; }

	void timing_ex(double *out4, uint64_t *segments, bool reset) override {
		if (!timing_on_) timing_on_ = true;
		(void)hipStreamSynchronize(stream_);
		drain_pairs();
		/* out: time-parallel kernel, block-loop kernel, mixer, analyze+finalize (ms) */
		out4[0] = acc_ms_[2]; out4[1] = acc_ms_[0]; out4[2] = acc_ms_[1]; out4[3] = acc_ms_[3];
		if (segments) *segments = acc_launches_;
		if (reset) { acc_ms_[0] = acc_ms_[1] = acc_ms_[2] = acc_ms_[3] = 0; acc_launches_ = 0; }
	}

	void debug_dump(const char *what, const SegmentDesc &seg) {
		(void)hipStreamSynchronize(stream_);
		uint32_t n = cfg_.op_count < 8 ? cfg_.op_count : 8;
		std::vector<DevOp> h(n);
		(void)hipMemcpy(h.data(), ops_.p, n * sizeof(DevOp), hipMemcpyDeviceToHost);
		fprintf(stderr, "[sau-amd] %s: seg len %u off %u voices %u slots %u\n", what, seg.len,
				seg.pcm_offset, seg.n_voices, seg.n_slots);
		for (uint32_t v = 0; v < seg.n_voices && v < 4; ++v)
			fprintf(stderr, "  voice %u: run_len %u plan %u+%u ops %u+%u\n", v, seg.voices[v].run_len,
					seg.voices[v].plan_ofs, seg.voices[v].plan_len, seg.voices[v].ops_ofs, seg.voices[v].nops);
		{
			uint32_t wc = 0;
			(void)hipMemcpy(&wc, work_count_.p, 4, hipMemcpyDeviceToHost);
			{ /* voices the time-parallel path did not finish */
				std::vector<FastInfo> all(seg.n_voices);
				(void)hipMemcpy(all.data(), finfo_.p, all.size() * sizeof(FastInfo), hipMemcpyDeviceToHost);
				uint32_t shown = 0;
				for (size_t v = 0; v < all.size() && shown < 16; ++v)
					if (all[v].bail || all[v].total < seg.voices[v].run_len) {
						uint32_t dbg = 0;
						(void)hipMemcpy(&dbg, repair_.p + v * FAST_REPAIR_WORDS + 1, 4, hipMemcpyDeviceToHost);
						fprintf(stderr, "  voice %zu: fast total %u of %u, bail %u, H %u, seq %u; last unresolved hold: lane p_min+%u row %u step %u group %u (mod 4096)\n", v, all[v].total,
								seg.voices[v].run_len, all[v].bail, all[v].H, all[v].seq, dbg >> 24, (dbg >> 20) & 15, (dbg >> 12) & 255, dbg & 0xfff);
						++shown;
					}
			}
			std::vector<VoiceOut> vi(seg.n_voices < 4 ? seg.n_voices : 4);
			std::vector<uint32_t> fd(vi.size());
			std::vector<FastInfo> fi(vi.size());
			(void)hipMemcpy(vi.data(), set_.vinfo.p, vi.size() * sizeof(VoiceOut), hipMemcpyDeviceToHost);
			(void)hipMemcpy(fd.data(), fdone_.p, fd.size() * 4, hipMemcpyDeviceToHost);
			(void)hipMemcpy(fi.data(), finfo_.p, fi.size() * sizeof(FastInfo), hipMemcpyDeviceToHost);
			float v8[8] = {0};
			(void)hipMemcpy(v8, set_.vout.p, sizeof v8, hipMemcpyDeviceToHost);
			fprintf(stderr, "  work_count %u block_grid %u; row0: %g %g %g %g %g %g\n", wc, block_grid_, v8[0], v8[1],
					v8[2], v8[3], v8[4], v8[5]);
			for (size_t v = 0; v < vi.size(); ++v)
				fprintf(stderr, "  vinfo %zu: pan %g has_pan %u valid %u prow %u | fast total %u H %u bail %u done %u\n", v,
						vi[v].pan_const, vi[v].has_pan, vi[v].valid_len, vi[v].pan_row, fi[v].total, fi[v].H,
						fi[v].bail, fd[v]);
		}
		for (uint32_t i = 0; i < n; ++i) {
			const DevOp &o = h[i];
			fprintf(stderr, "  op %u: time %u flags %#x type %u wave %u phase %u prev_s %g fb %g\n", i, o.time,
					o.flags, o.type, o.wave, o.phase, o.prev_s, o.fb_s);
			for (int l = 0; l < 6; ++l)
				if (o.line[l].flags || o.line[l].v0 != 0.f)
					fprintf(stderr, "     line %d: v0 %g vt %g pos %u end %u type %u flags %#x\n", l,
							o.line[l].v0, o.line[l].vt, o.line[l].pos, o.line[l].end, o.line[l].type,
							o.line[l].flags);
		}
	}

private:
	/* kind: 0 block-loop kernel, 1 mixer, 2 time-parallel kernel, 3 analyze/finalize */
	struct TimedPair { hipEvent_t a, b; int kind; bool used; };
	TimedPair *new_pair(int kind) {
		if (timing_level_ == 1 && kind != 2) return nullptr; /* level 1: dominant kernel only */
		if (n_used_ == events_.size()) {
			if (events_.size() >= 4096) { (void)hipStreamSynchronize(stream_); drain_pairs(); }
			else {
				TimedPair p; p.kind = 0; p.used = false;
				if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
				events_.push_back(p);
			}
		}
		TimedPair *p = &events_[n_used_++];
		p->kind = kind; p->used = true;
		return p;
	}
	void drain_pairs() {
		for (size_t i = 0; i < n_used_; ++i) {
			float ms = 0;
			if (hipEventElapsedTime(&ms, events_[i].a, events_[i].b) == hipSuccess) {
				acc_ms_[events_[i].kind & 3] += ms;
			}
		}
		n_used_ = 0;
	}

	BackendConfig cfg_;
	hipStream_t stream_ = nullptr;
	size_t lds_limit_ = 64 * 1024;
	int geo_ = 0;
	bool debug_ = false;
	uint32_t row_stride_ = 0;
	size_t pcm_row_ = 0;
	WaveConst wconst_[12];
	DevBuf<DevOp> ops_;
	DevBuf<DevOp> ops_snap_[4]; /* save_state() */
	DevBuf<Step> steps_;
	DevBuf<FastIds> fast_ids_; /* [2][n_steps_total_]: without / with frequency blocks */
	uint32_t n_steps_total_ = 0;
	DevBuf<uint32_t> op_ids_;
	DevBuf<VoiceDesc> voices_;
	/* what the mixer reads: voice rows, pan rows, per-row records, stream records */
	struct MixSet {
		DevBuf<float> vout, pan;
		DevBuf<VoiceOut> vinfo;
		DevBuf<MixStream> mstreams;
		uint32_t vout_rows = 0;
		std::vector<MixStream> ms_sent;
		const void *ms_dev = nullptr;
	};
	MixSet set_;
	DevBuf<int16_t> pcm_;
	DevBuf<OpUpdate> recs_;
	const TableSet *tables_ = nullptr;
	PinBuf<int16_t> h_pcm_; /* fetch_pcm() staging */
	PinBuf<unsigned char> arena_; /* stage() */
	size_t arena_used_ = 0;
	std::vector<VoiceDesc> voices_sent_;   /* what the device copies hold */
	std::vector<MixStream> ms_host_;
	const void *voices_dev_ = nullptr;
	std::deque<TimedPair> events_; /* (a deque: pairs handed out stay where they are while more are added) */
	size_t n_used_ = 0;
	bool timing_on_ = false;
	double acc_ms_[4] = {0, 0, 0, 0};
	uint64_t acc_launches_ = 0;
	bool fast_enabled_ = true;
	int timing_level_ = 2;
	DevBuf<FastInfo> finfo_;
	DevBuf<uint32_t> fdone_, worklist_, work_count_;
	DevBuf<float> big_slots_;     /* block buffers of render_kernel<1, 1, 1, true>: [workgroup][slot][64] */
	DevBuf<unsigned char> fsteps_, flines_, faux_;
	DevBuf<unsigned long long> scan_;
	DevBuf<uint32_t> pass_flags_, repair_;
	hipEvent_t fetch_ev_[4] = {nullptr, nullptr, nullptr, nullptr}; /* per host slot of the drop-in generator's read-ahead */
	int dev_ = 0;
	int want_dev_ = -1; /* the device the creator asked for (-1: SAU_AMD_DEVICE, else 0) */
	std::map<void *, size_t> host_blocks_; /* alloc_host() blocks and their pool sizes */
	uint32_t multi_min_ = 256;
	uint32_t multi_teams_ = 16; /* SAU_AMD_MULTI_TEAMS: 8 or 16 single-wave teams per workgroup */
	uint32_t fast_rows_ = 8;
	static inline size_t chain_lds_configured_[16] = {}; /* chain_kernel's LDS attribute per device (two launch sites) */
	bool seq_enabled_ = true, two_pass_enabled_ = true, chain_enabled_ = true, chain_inline_ = false, chain_early_ = true;
	uint32_t chain_chunks_ = 0;           /* SAU_AMD_CHAIN_CHUNKS: a fixed number of chunks per segment (0: by frames) */
	uint32_t chain_chunk_frames_ = 16384; /* SAU_AMD_CHAIN_CHUNK_FRAMES */
	bool inc_rows_enabled_ = true;
	bool lookback_enabled_ = true;
	uint32_t look_min_voices_ = 1; /* SAU_AMD_LOOK_MIN_VOICES: segments with fewer voices keep the several-pass form */
	uint32_t look_rows_ = 8; /* rows per pass of the single-pass build (SAU_AMD_LOOK_ROWS; 0: no such build, the full one takes every voice) */
	DevBuf<unsigned long long> look_;
	uint32_t look_epoch_ = 0;
	DevBuf<uint32_t> inc_rows_;
	hipStream_t chain_stream_ = nullptr;
	std::vector<hipEvent_t> chain_ev_;
	DevBuf<float> chain_rows_;
	bool chain_rows_failed_once_ = false;
	unsigned chain_rows_failures_ = 0; /* allocations of rows that failed on this backend */
	size_t free_hint_ = 0;             /* bytes free on the device when the process first opened it */
	DevBuf<ChainDesc> chain_desc_;
	DevBuf<unsigned char> fplines_;
	uint32_t block_grid_ = 1;
	uint32_t fk_grid_ = FK_GRID;
	bool dyn_enabled_ = true, lean_enabled_ = true, mix_few_enabled_ = true;
	bool short_last_chunk_ = true; /* SAU_AMD_NO_SHORT_LAST_CHUNK */
	bool early_mix_enabled_ = true; /* segments with chains in chunks: the mixer follows the final passes chunk by chunk */
	hipEvent_t after_ev_ = nullptr; /* order_after() */
	bool after_pending_ = false;
	uint32_t early_mixed_blocks_ = 0; /* 256-frame blocks of this segment that early launches have mixed */
	bool eight_xcds_ = true;          /* the device is a whole MI355X (init) */
	bool xcd_queues_ = true;          /* a closed-form launch's tasks come from one queue per XCD, chunk-major (SAU_AMD_NO_XCD_QUEUES: the one counter, voice-major) */
	bool inmix_enabled_ = true;       /* a many-voice stream's closed-form launch mixes its own rows (SAU_AMD_NO_INMIX: the mixer alone) */
	uint32_t inmix_min_voices_ = 64;  /* ... from that many voices on (SAU_AMD_INMIX_MIN_VOICES) */
	uint32_t inmix_min_steps_ = 2;    /* ... of two steps or more (SAU_AMD_INMIX_MIN_STEPS): behind one-step voices (config 2) the launch's mixing
	                                   * costs it more than the mixer saves -- step 0.269-0.272 ms with it, 0.258-0.261 without */
	bool inmix_live_ = false;         /* this segment's launch did: mix_kernel looks at the control words */
	bool inmix_taper_ = false;        /* SAU_AMD_INMIX_TAPER: the queues' last eight chunks are short ones, so that less is left to mix_kernel. Measured
	                                   * slower on config 3 (2.04 -> 2.08 ms per step, the launch 1.91 -> 1.94: profiles/r06_ab.txt): off */
	uint32_t inmix_at_ = 13;           /* which of a chunk's tasks mix the chunk before: from this many sixteenths into it (SAU_AMD_INMIX_AT) */
	bool inmix_report_ = false;       /* SAU_AMD_INMIX_REPORT: a line on stderr per such segment (tests) */
	DevBuf<uint32_t> inmix_ctl_;
	DevBuf<uint32_t> tail_ok_;        /* [stream]: the look-back launch mixes this stream itself (k_fast_types.h: FastParams.tail_ok) */
	bool tailmix_enabled_ = false;    /* SAU_AMD_TAILMIX: the look-back launch mixes few-voice streams itself. Exact (357 GPU tests with it on), and
	                                   * measured slower: BASELINE config 4 4.19 -> 4.26 ms per step -- the launch is issue-bound and pays 0.32 ms for
	                                   * the six flops, the clamp and the conversion per frame (and 42 spilled registers), the mixer gives back 0.25
	                                   * (profiles/r06_ab.txt). Off; kept for the record */
	bool tail_live_ = false;          /* this segment's look-back launch did: mix_few_kernel looks at tail_ok_ */
	bool mix64_enabled_ = true;       /* behind a launch that has mixed most tiles itself: the mixer in blocks of 64 frames (SAU_AMD_NO_MIX64: 256) */
	bool duo_enabled_ = true;         /* closed-form and look-back voices of a segment in one launch where there are plenty of both (SAU_AMD_NO_DUO: two launches) */
	bool wide_tabs_ = true;
	uint32_t more_rows_ = 12;
	uint32_t lean_rows_ = 6;    /* SAU_AMD_LEAN_ROWS: rows per pass of fast_kernel<T, 3> at most (6, 5 or 4) */
	/* A chain kernel's workgroup is three waves on a latency-bound recurrence. SAU_AMD_CHAIN_ALONE=1 (a tuning switch): while a
	 * launch has no more workgroups than the device has CUs, each asks for more than half a CU's LDS and so gets a CU of its own.
	 * Measured without effect -- R feedback banks of 8 / 64 / 1024 / 4096 voices 521 571 501 427 ns per frame with it, 486-511
	 * 574-609 484-501 426 without, config 5 45.96-46.06 against 45.77-46.00 ms: the workgroups do not slow each other; what varies
	 * from run to run on a nearly idle device (one voice: 324-418 ns) is not the placement. Off. */
	size_t chain_alone_lds(size_t need, uint32_t grid) const {
		if (!chain_alone_ || grid > cus_ || need * 2 > lds_limit_) return need;
		const size_t half = lds_limit_ / 2 + 1024;
		return need > half ? need : half;
	}
	bool chain_alone_ = false;
	bool inner_enabled_ = true;  /* the 12-row wide closed-form build as two launches, edges and inner groups (SAU_AMD_NO_INNER: one) */
	uint32_t cus_ = 256;
	uint32_t dyn_groups_ = 12;  /* row groups per task of a closed-form launch, at least (SAU_AMD_DYN_GROUPS) */
	/* ... and steps x row groups per task, at least (SAU_AMD_DYN_FLOOR), and the tasks per wave from which the queues deal them out
	 * (fewer: a fixed share per wave; SAU_AMD_DYN_MIN_TASKS). Round 6: 48 -> 24 and 4 -> 2, which takes BASELINE config 2 (256 voices of
	 * one step: 20 tasks a voice were 1.25 a wave, so every wave got a sixteenth of a voice, fixed) to the queues: its launch 0.192 ->
	 * 0.163-0.166 ms, the step 0.287 -> 0.259 (floors 4-12 the same, 16: 0.272; profiles/r06_ab.txt [17]) */
	uint32_t dyn_floor_ = 24;
	uint32_t dyn_min_tasks_ = 2;
	bool look_words_real_ = false; /* this segment's look-back words in HBM are usable (not the token block) */
	DevBuf<uint32_t> vlists_;   /* [2][n_voices]: analyze_kernel's lists of closed-form and look-back voices (split launches) */
	uint32_t look_wpv_ = 1;     /* this segment's single-pass launch: waves per voice, and whether every voice sits */
	bool look_inside_ = true;   /* inside one workgroup (LDS rings, no waits across workgroups) */
};

HipBackend *create_hip_backend(std::string &err, int device) {
	if (device_count() <= 0) {
		err = "no HIP device available (this backend has no CPU fallback)";
		return nullptr;
	}
	return new HipBackendImpl(device);
}

} /* namespace sauhip */

