/* hip_backend.hip -- gfx950 kernels and the HIP backend of the generator.
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
#include <vector>

namespace sauhip {

using namespace saudev;
using sauengine::BackendConfig;
using sauengine::SegmentDesc;

/* ------------------------------------------------------------------------ */
/* device side                                                              */
/* ------------------------------------------------------------------------ */

typedef uint32_t __attribute__((may_alias)) u32_alias; /* raw copies of typed structs */

struct RenderParams {
	const VoiceDesc *voices;
	const Step *steps;
	const uint32_t *op_ids;
	DevOp *ops;
	float *vout;           /* [row][row_stride] carrier blocks */
	float *pan;            /* [pan row][row_stride] */
	VoiceOut *vinfo;       /* [row] */
	const HerpC23 *g_c23;  /* [12][2048] */
	const HerpC01 *g_c01;
	uint32_t row_stride;
	uint32_t seg_len;
	uint32_t n_slots;
	uint32_t n_main;       /* main-pool slots (slot_index() base) */
	uint32_t max_ops;
	uint32_t max_steps;    /* longest plan of the launch (LDS copy) */
	const uint32_t *fast_done; /* [voice row] frames already rendered by fast_kernel */
	const uint32_t *worklist;  /* voice rows that still need the block loop */
	const uint32_t *work_count;
	uint32_t n_tabs;       /* wave types staged in LDS */
	uint32_t team_bytes;   /* LDS bytes per team (several teams per workgroup only) */
	int8_t tab_of_wave[12];/* LDS table index per wave id, or -1 */
	uint8_t wave_of_tab[12];
	WaveConst wc[12];
};

struct Misc {
	WaveConst wc[12];      /* per-wave constants, copied from the launch parameters */
	int32_t tab_of_wave[12];
	uint16_t len_stack[MAX_NEST + 1]; /* block lengths per nesting level (<= 1024 each) */
	uint32_t tot32[16];
	unsigned long long tot64[16];
	uint32_t flag;
	/* time-parallel regime */
	uint32_t fast_bad, min_time, bail, fast_depth;
	uint32_t pad;
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t t = __shfl_up(v, d);
		if (lane >= d) v += t;
	}
	return v;
}
__device__ __forceinline__ unsigned long long wave_incl_scan64(unsigned long long v, int lane) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		unsigned long long t = __shfl_up(v, d);
		if (lane >= d) v += t;
	}
	return v;
}

/* Values that are the same in every lane of the workgroup (plan steps,
 * operator state, block lengths) are loaded from LDS into vector registers;
 * moving them to scalar registers lets the compiler use scalar branches and
 * scalar arithmetic for all the per-step bookkeeping. */
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return bits_f(uni(f_bits(v))); }
__device__ __forceinline__ bool uni(bool v) { return uni((uint32_t)v) != 0; }
__device__ __forceinline__ double uni(double v) {
	union { double d; uint32_t u[2]; } c; c.d = v;
	c.u[0] = uni(c.u[0]); c.u[1] = uni(c.u[1]);
	return c.d;
}
__device__ __forceinline__ LineState uni(const LineState &l) {
	LineState r;
	r.v0 = uni(l.v0); r.vt = uni(l.vt); r.pos = uni(l.pos); r.end = uni(l.end);
	r.type = uni(l.type); r.flags = uni(l.flags);
	return r;
}
__device__ __forceinline__ Step uni(const Step &st) {
	union { Step s; uint32_t u[4]; } c; c.s = st;
	c.u[0] = uni(c.u[0]); c.u[1] = uni(c.u[1]); c.u[2] = uni(c.u[2]); c.u[3] = uni(c.u[3]);
	return c.s;
}

/* inclusive sum over the 64 lanes with DPP moves (no LDS): four shifts inside
 * each row of 16, then the rows' totals passed on with row_bcast 15 and 31 */
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
#define SAU_DPP_ADD(ctrl, rmask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false)
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
	SAU_DPP_ADD(0x142 /* row_bcast:15 */, 0xa);
	SAU_DPP_ADD(0x143 /* row_bcast:31 */, 0xc);
#undef SAU_DPP_ADD
	return v;
}

__device__ __forceinline__ unsigned long long wave_incl_scan64_dpp(unsigned long long v) {
#define SAU_DPP_ADD64(ctrl, rmask, bc) do { \
		const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, ctrl, rmask, 0xf, bc); \
		const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), ctrl, rmask, 0xf, bc); \
		v += ((unsigned long long)hi_ << 32) | lo_; } while (0)
	SAU_DPP_ADD64(0x111, 0xf, true);
	SAU_DPP_ADD64(0x112, 0xf, true);
	SAU_DPP_ADD64(0x114, 0xf, true);
	SAU_DPP_ADD64(0x118, 0xf, true);
	SAU_DPP_ADD64(0x142, 0xa, false);
	SAU_DPP_ADD64(0x143, 0xc, false);
#undef SAU_DPP_ADD64
	return v;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
	return ((unsigned long long)hi << 32) | lo;
}

/* Decoupled look-back over a voice's row groups (FastParams.look). The calling wave owns group cg and its
 * group total `tot`; returns the sum of all earlier groups' totals. Words carry value and status together,
 * so one relaxed device-scope store publishes and one load observes -- no fences. A wave only ever waits
 * for groups before its own: those belong to waves of this launch that are resident (the grid is at most
 * one workgroup per CU) or to an earlier launch of the same segment. */
constexpr uint32_t LOOK_AGG = 1, LOOK_PREFIX = 2;
/* The same words in LDS, for a voice whose waves all sit in one workgroup (2, 4, 8 or 16 of them): a ring of
 * 4 x waves entries per oscillator, tagged with the group's number + 1 (LDS starts out zeroed). A wave that
 * writes group g has finished group g - waves, so every wave of the voice has published at least up to round
 * r - 2 and reads no further back than its own prefix of round r - 3: the entry of g - 4 x waves is dead. */
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
__device__ __forceinline__ unsigned long long look_word(uint32_t tag, uint32_t status, uint32_t value) {
	return ((unsigned long long)((tag << 2) | status) << 32) | value;
}
template <bool LDS> __device__ __forceinline__ void look_store(unsigned long long *p, unsigned long long w) {
	if (LDS) __hip_atomic_store((lds_u64 *)p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	else __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LDS> __device__ __forceinline__ unsigned long long look_load(unsigned long long *p) {
	if (LDS) return __hip_atomic_load((lds_u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* LDS: entry of group i at ent[i & (ring - 1)], tag i + 1; HBM: at ent[i], tag = the segment's epoch */
template <bool LDS> __device__ __forceinline__ uint32_t lookback32(unsigned long long *ent, const uint32_t cg, const uint32_t tot,
		const uint32_t epoch, const uint32_t ring, const int l) {
	auto at = [&](uint32_t i) { return LDS ? (i & (ring - 1)) : i; };
	auto tag = [&](uint32_t i) { return LDS ? i + 1 : epoch; };
	if (cg == 0) {
		if (l == 0) look_store<LDS>(&ent[0], look_word(tag(0), LOOK_PREFIX, tot));
		return 0;
	}
	if (l == 0) look_store<LDS>(&ent[at(cg)], look_word(tag(cg), LOOK_AGG, tot));
	uint32_t excl = 0;
	int p = (int)cg - 1; /* the nearest group not yet accounted for */
	for (;;) {
		const int idx = p - l; /* lane l looks at the group l before it; before group 0 the prefix is 0 */
		unsigned long long e = 0;
		if (idx >= 0) e = look_load<LDS>(&ent[at((uint32_t)idx)]);
		const uint32_t hi = (uint32_t)(e >> 32);
		const uint32_t st = idx < 0 ? LOOK_PREFIX : (hi >> 2) == tag((uint32_t)idx) ? (hi & 3u) : 0u;
		const unsigned long long m_pref = __ballot(st == LOOK_PREFIX), m_none = __ballot(st == 0);
		const int first_pref = m_pref ? __builtin_ctzll(m_pref) : 64;
		const int first_none = m_none ? __builtin_ctzll(m_none) : 64;
		const int upto = first_pref < first_none ? first_pref + 1 : first_none; /* lanes [0, upto) count */
		const uint32_t part = wave_incl_scan_dpp(l < upto && idx >= 0 ? (uint32_t)e : 0u);
		excl += (uint32_t)__builtin_amdgcn_readlane((int)part, 63);
		if (first_pref < first_none) break;
		p -= upto;
		if (upto == 0) __builtin_amdgcn_s_sleep(LDS ? 1 : 2);
	}
	if (l == 0) look_store<LDS>(&ent[at(cg)], look_word(tag(cg), LOOK_PREFIX, excl + tot));
	return excl;
}
/* 64-bit totals (R oscillators' cycle counters): low and high halves in two arrays, a pair counts once both
 * words show the same status */
template <bool LDS> __device__ __forceinline__ unsigned long long lookback64(unsigned long long *ent_lo, unsigned long long *ent_hi,
		const uint32_t cg, const unsigned long long tot, const uint32_t epoch, const uint32_t ring, const int l) {
	auto at = [&](uint32_t i) { return LDS ? (i & (ring - 1)) : i; };
	auto tag = [&](uint32_t i) { return LDS ? i + 1 : epoch; };
	auto publish = [&](uint32_t i, uint32_t status, unsigned long long v) {
		look_store<LDS>(&ent_lo[at(i)], look_word(tag(i), status, (uint32_t)v));
		look_store<LDS>(&ent_hi[at(i)], look_word(tag(i), status, (uint32_t)(v >> 32)));
	};
	if (cg == 0) {
		if (l == 0) publish(0, LOOK_PREFIX, tot);
		return 0;
	}
	if (l == 0) publish(cg, LOOK_AGG, tot);
	unsigned long long excl = 0;
	int p = (int)cg - 1;
	for (;;) {
		const int idx = p - l;
		unsigned long long a = 0, b = 0;
		if (idx >= 0) {
			a = look_load<LDS>(&ent_lo[at((uint32_t)idx)]);
			b = look_load<LDS>(&ent_hi[at((uint32_t)idx)]);
		}
		const uint32_t ha = (uint32_t)(a >> 32), hb = (uint32_t)(b >> 32);
		const uint32_t st = idx < 0 ? LOOK_PREFIX : ((ha >> 2) == tag((uint32_t)idx) && ha == hb) ? (ha & 3u) : 0u;
		const unsigned long long m_pref = __ballot(st == LOOK_PREFIX), m_none = __ballot(st == 0);
		const int first_pref = m_pref ? __builtin_ctzll(m_pref) : 64;
		const int first_none = m_none ? __builtin_ctzll(m_none) : 64;
		const int upto = first_pref < first_none ? first_pref + 1 : first_none;
		const unsigned long long v = ((unsigned long long)(uint32_t)b << 32) | (uint32_t)a;
		const unsigned long long part = wave_incl_scan64_dpp(l < upto && idx >= 0 ? v : 0ull);
		excl += readlane64(part, 63);
		if (first_pref < first_none) break;
		p -= upto;
		if (upto == 0) __builtin_amdgcn_s_sleep(LDS ? 1 : 2);
	}
	if (l == 0) publish(cg, LOOK_PREFIX, excl + tot);
	return excl;
}

/* rint(p * 2^31) wrapped to 32 bits for |p| < 2^20: in f64, p + 1.5 * 2^21
 * has an ulp of 2^-31, so the addition rounds p to a multiple of 2^-31
 * (nearest-even, as llrintf does in the default mode) and leaves that
 * multiple, mod 2^32, in the low word of the significand. */
__device__ __forceinline__ uint32_t rint32w_p31_small(float p) {
	return (uint32_t)__double2loint((double)p + 0x1.8p21);
}

/* Where the coefficient tables of one wave type are read from. */
struct TabRef {
	const HerpC23 *c23;
	const HerpC01 *c01;
	bool in_lds;
};

/* LDS copies are read through LDS-typed pointers: a pointer that may be either
 * kind compiles to flat loads, which cost several times a ds_read and, in the
 * serial feedback loops, sat on the critical path of every sample. */
typedef const double __attribute__((address_space(3))) *lds_f64_ptr;
typedef const float __attribute__((address_space(3))) *lds_f32_ptr;
__device__ __forceinline__ double herp_lookup(const TabRef &t, uint32_t phase) {
	uint32_t ind = phase >> SLEN_BITS;
	HerpC23 hi;
	HerpC01 lo;
	if (t.in_lds) {
		lds_f64_ptr p23 = (lds_f64_ptr)(const double *)(t.c23 + ind);
		lds_f32_ptr p01 = (lds_f32_ptr)(const float *)(t.c01 + ind);
		hi.c3 = p23[0]; hi.c2 = p23[1];
		lo.c1 = p01[0]; lo.c0 = p01[1];
	} else {
		hi = t.c23[ind];
		lo = t.c01[ind];
	}
	return herp_poly(hi, lo, phase);
}

/* One wave type's tables as LDS addresses (32-bit) or global pointers. */
template <bool LDS> struct TabAt;
typedef const HerpC23 __attribute__((address_space(3))) *lds_c23_ptr;
typedef const HerpC01 __attribute__((address_space(3))) *lds_c01_ptr;
template <> struct TabAt<true> {
	lds_c23_ptr c23; lds_c01_ptr c01;
	__device__ __forceinline__ explicit TabAt(const TabRef &t)
		: c23((lds_c23_ptr)t.c23), c01((lds_c01_ptr)t.c01) {}
	__device__ __forceinline__ double lookup(uint32_t phase) const {
		const uint32_t ind = phase >> SLEN_BITS;
		HerpC23 hi; HerpC01 lo;
		hi.c3 = c23[ind].c3; hi.c2 = c23[ind].c2;
		lo.c1 = c01[ind].c1; lo.c0 = c01[ind].c0;
		return herp_poly(hi, lo, phase);
	}
};
template <> struct TabAt<false> {
	const HerpC23 *c23; const HerpC01 *c01;
	__device__ __forceinline__ explicit TabAt(const TabRef &t) : c23(t.c23), c01(t.c01) {}
	__device__ __forceinline__ double lookup(uint32_t phase) const {
		const uint32_t ind = phase >> SLEN_BITS;
		return herp_poly(c23[ind], c01[ind], phase);
	}
};

/* Carried state of one W oscillator in its feedback loop. */
struct SelfmodState {
	uint32_t prev_phase;
	double prev_Is;
	float prev_s, fb_s;
};

/* wosc.h:273-310 for one block, one lane: the loop carries only
 * fb_s -> phase -> table -> sample. Base phases and self-modulation amounts
 * were laid out in LDS by the whole wave; entry e of sample j is j + 1 plus
 * one skipped (halo) entry per `span` samples. The next sample's inputs are
 * fetched while the current one is computed; a repeated phase holds the
 * previous sample (wosc.h:292-293), decided by selects, not by a branch. */
template <bool LDS, int SPAN /* samples per wave span when several waves share a block, else 0 */>
__device__ __forceinline__ void selfmod_serial(const TabRef &tab, SelfmodState &st, const WaveConst &wc,
		const float *pmaS, u32_alias *baseS /* also receives the samples */, uint32_t len) {
	typedef uint32_t __attribute__((address_space(3))) *lds_u32_w;
	const TabAt<LDS> at(tab);
	uint32_t prev_phase = st.prev_phase;
	double prev_Is = st.prev_Is;
	float prev_s = st.prev_s, fb_s = st.fb_s;
	lds_f32_ptr pm = (lds_f32_ptr)pmaS + 1; /* entry of the current sample */
	lds_u32_w bs = (lds_u32_w)(uint32_t *)baseS + 1;
	uint32_t r = 0;
	float pma_n = *pm;
	uint32_t base_n = *bs;
	for (uint32_t j = 0; j < len; ++j) {
		const float pma = pma_n;
		const uint32_t base = base_n;
		const lds_u32_w cur = bs;
		++pm; ++bs;
		if (SPAN && ++r == (uint32_t)SPAN) { r = 0; ++pm; ++bs; } /* skip the next wave's halo entry */
		if (j + 1 < len) { pma_n = *pm; base_n = *bs; }
		const float p = fb_s * pma;
		uint32_t ofs = rint32w_p31_small(p);
		if (__builtin_expect(!(fabsf(p) < 0x1p20f), 0)) ofs = rint32w(p * 0x1p31f);
		const uint32_t phase = base + ofs;
		const int32_t d = (int32_t)(phase - prev_phase);
		const double Isv = at.lookup(phase);
		const float sv_new = wosc_diff(Isv, prev_Is, d, wc.diff_scale, wc.diff_offset);
		const bool hold = d == 0;
		const float sv = hold ? prev_s : sv_new;
		prev_Is = hold ? prev_Is : Isv;
		prev_phase = phase; /* equal to the old one when held */
		prev_s = sv;
		*cur = f_bits(sv);
		fb_s = (fb_s + sv) * 0.5f;
	}
	st.prev_phase = prev_phase; st.prev_Is = prev_Is; st.prev_s = prev_s; st.fb_s = fb_s;
}

template <int W, int T>
struct Geo {
	static constexpr int NP = 64 * T;        /* slot entries per wave */
	static constexpr int NB = W * (NP - 1);  /* new samples per block */
	static constexpr int SLOT = W * NP;      /* floats per slot */
};

/* entry index of sample j (>= 0) */
template <int W, int T>
__device__ __forceinline__ uint32_t entry_of(uint32_t j) {
	constexpr uint32_t S = Geo<W, T>::NP - 1;
	uint32_t w = j / S;
	return w * Geo<W, T>::NP + (j - w * S) + 1;
}

/* store one owned sample, keeping the next wave's halo copy in step */
template <int W, int T>
__device__ __forceinline__ void slot_put(float *slot, int w, int p, float v) {
	slot[w * Geo<W, T>::NP + p] = v;
	if (p == Geo<W, T>::NP - 1 && w + 1 < W)
		slot[(w + 1) * Geo<W, T>::NP] = v;
}

/* Frequency of one operator for a block when it is a single value:
 * a held line (no sweep pending) that is absolute, or a ratio of a parent
 * frequency that is itself a single value. */
__device__ __forceinline__ bool const_freq(const LineState &ls, bool has_mul, bool parent_const,
		float parent_f, float &fc) {
	if (ls.flags & LP_GOAL) return false;
	if (has_mul && (ls.flags & LP_STATE_RATIO)) {
		if (!parent_const) return false;
		fc = ls.v0 * parent_f; /* sau/line.c:72 v0 * mulbuf[i] */
		return true;
	}
	fc = ls.v0;
	return true;
}

/* A team is the W waves that render one voice. V == 1: the workgroup is one
 * team and its steps are separated by workgroup barriers. V > 1 (W == 1):
 * every wave of the workgroup is a team of its own with a private LDS area --
 * wave-synchronous, no barriers -- so that a CU keeps V voices in flight;
 * that is what the serial feedback recurrences need (one lane per voice). */
template <int V>
__device__ __forceinline__ void team_sync() {
	if constexpr (V == 1) __syncthreads();
	else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int W, int T, int V>
__global__ void __launch_bounds__(64 * W * V) render_kernel(RenderParams P) {
	static_assert(V == 1 || W == 1, "several teams per workgroup are single waves");
	using G = Geo<W, T>;
	constexpr int NTHREADS = 64 * W * V;
	extern __shared__ __align__(16) unsigned char lds[];
	const int team = V > 1 ? (int)uni((uint32_t)threadIdx.x >> 6) : 0;
	const int tid = V > 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x; /* within the team */
	const int w = tid >> 6;
	const int l = tid & 63;

	HerpC23 *t23 = (HerpC23 *)lds;
	HerpC01 *t01 = (HerpC01 *)(lds + (size_t)P.n_tabs * WAVE_LEN * sizeof(HerpC23));
	float *slots = (float *)(lds + (size_t)P.n_tabs * WAVE_LEN * (sizeof(HerpC23) + sizeof(HerpC01)) +
			(size_t)team * P.team_bytes);
	DevOp *ops = (DevOp *)(slots + (size_t)P.n_slots * G::SLOT);
	Misc *misc = (Misc *)(ops + P.max_ops);
	Step *plan = (Step *)(misc + 1); /* this voice's steps, read every block */

	const uint32_t n_work = *P.work_count;
	if (blockIdx.x * V >= n_work) return;

	/* stage coefficient tables (16-byte copies) */
	for (uint32_t t = 0; t < P.n_tabs; ++t) {
		const uint32_t wave = P.wave_of_tab[t];
		const uint4 *s23 = (const uint4 *)(P.g_c23 + (size_t)wave * WAVE_LEN);
		uint4 *d23 = (uint4 *)(t23 + (size_t)t * WAVE_LEN);
		for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += NTHREADS) d23[i] = s23[i];
		const uint2 *s01 = (const uint2 *)(P.g_c01 + (size_t)wave * WAVE_LEN);
		uint2 *d01 = (uint2 *)(t01 + (size_t)t * WAVE_LEN);
		for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += NTHREADS) d01[i] = s01[i];
	}
	if (V > 1) __syncthreads(); /* tables are shared by the teams; nothing else is */

	/* persistent over the work list: voices the time-parallel path finished
	 * never get here */
	for (uint32_t item = blockIdx.x * V + team; item < n_work; item += gridDim.x * V) {
	const uint32_t vrow_id = P.worklist[item];
	const VoiceDesc vd = P.voices[vrow_id];
	const uint32_t *my_ids = P.op_ids + vd.ops_ofs;
	float *vrow = P.vout + (size_t)vd.out_row * P.row_stride;
	float *prow = (vd.pan_dynamic_row != ~0u) ? P.pan + (size_t)vd.pan_dynamic_row * P.row_stride : nullptr;
	uint32_t done = P.fast_done[vrow_id]; /* frames rendered by fast_kernel */
	uint32_t produced = done;
	Lattice lat;
	lat.e0 = uni(vd.lat.e0); lat.span_left = uni(vd.lat.span_left); lat.call_len = uni(vd.lat.call_len);
	team_sync<V>(); /* previous voice's LDS contents are no longer needed */
	for (uint32_t i = tid; i < vd.nops * 64; i += 64 * W)
		((u32_alias *)ops)[i] = ((const u32_alias *)&P.ops[my_ids[i >> 6]])[i & 63];
	{
		const u32_alias *src = (const u32_alias *)(P.steps + vd.plan_ofs);
		for (uint32_t i = tid; i < vd.plan_len * 4; i += 64 * W) ((u32_alias *)plan)[i] = src[i];
	}
	if (tid < 12) {
		misc->wc[tid] = P.wc[tid];
		misc->tab_of_wave[tid] = P.tab_of_wave[tid];
	}
	if (tid == 0) misc->flag = 0;
	team_sync<V>();

	/* per-thread sample geometry: p = l*T + k, block sample j = w*(NP-1) + p - 1 */
	const int p0 = l * T;
	const int jbase = w * (G::NP - 1) + p0 - 1;

	while (done < vd.run_len) {
		if (uni(ops[vd.carr_local].time) == 0) break; /* generator.c:839 */
		const uint32_t blen = min((uint32_t)G::NB, vd.run_len - done);
		uint32_t depth = 0;
		uint32_t cur_len = blen;
		bool block_ended = false;

		for (uint32_t si = 0; si < vd.plan_len && !block_ended; ++si) {
			Step st = uni(plan[si]);
			{ /* slot ids -> memory indices (two pools, sau_dev_types.h) */
				const uint32_t nm = P.n_main;
				if (st.out != NO_SLOT) st.out = (uint8_t)slot_index(st.out, nm);
				if (st.freq != NO_SLOT) st.freq = (uint8_t)slot_index(st.freq, nm);
				if (st.fmul != NO_SLOT) st.fmul = (uint8_t)slot_index(st.fmul, nm);
				if (st.pm != NO_SLOT) st.pm = (uint8_t)slot_index(st.pm, nm);
				if (st.fpm != NO_SLOT) st.fpm = (uint8_t)slot_index(st.fpm, nm);
				if (st.amp != NO_SLOT) st.amp = (uint8_t)slot_index(st.amp, nm);
				if (st.sm != NO_SLOT) st.sm = (uint8_t)slot_index(st.sm, nm);
				if (st.kind == ST_OSC && st.tmp != NO_SLOT) st.tmp = (uint8_t)slot_index(st.tmp, nm);
			}
			const uint32_t parent_len = cur_len;
			DevOp *op = &ops[st.op];
			const uint32_t op_flags = uni(op->flags);
			if (st.flags & SF_BEGIN) { /* generator.c:694-698 */
				if (tid == 0) misc->len_stack[depth] = (uint16_t)cur_len;
				++depth;
				const uint32_t op_time = uni(op->time);
				if (!(op_flags & OPF_TIME_INF) && op_time < cur_len) cur_len = op_time;
			}
			const uint32_t len = cur_len;
			bool owned[T];
#pragma unroll
			for (int k = 0; k < T; ++k) owned[k] = (p0 + k >= 1) && (jbase + k < (int)len);

			/* Every step: reads of operator state and of input slots come first,
			 * then barrier A, then slot stores and state write-backs, then
			 * barrier B (steps that exchange data add barriers in between). */
			switch (st.kind) {
			case ST_ZERO: {
				float *out = slots + (size_t)st.out * G::SLOT;
				team_sync<V>();
#pragma unroll
				for (int k = 0; k < T; ++k)
					if (owned[k]) slot_put<W, T>(out, w, p0 + k, 0.f);
				break;
			}
			case ST_LINE: {
				float *out = slots + (size_t)st.out * G::SLOT;
				const float *mul = st.fmul != NO_SLOT ? slots + (size_t)st.fmul * G::SLOT : nullptr;
				LineState ls = uni(op->line[st.which]);
				/* a held frequency is passed on as one value instead of a block */
				bool pconst = false; float pf = 0.f;
				if (mul && st.prov != NO_SLOT) { pconst = uni(ops[st.prov].rt_fconst_valid) != 0; pf = uni(ops[st.prov].rt_fconst); }
				float fc = 0.f;
				const bool lazy = st.which == L_FREQ && !(st.flags & SF_FORCE) && st.op < 255 &&
					const_freq(ls, mul != nullptr, pconst, pf, fc);
				float v[T];
				if (!lazy) {
					/* the provider's block may be a single value (never stored) */
					const bool mconst = mul && pconst;
					const float m0 = mul ? (mconst ? pf : mul[1]) : 0.f;
					LineBlock lb = line_block_v(ls, len, mul != nullptr, m0);
					line_begin_state(ls, len, mul != nullptr, m0, lat, done);
#pragma unroll
					for (int k = 0; k < T; ++k) {
						if (owned[k]) {
							float m = mul ? (mconst ? pf : mul[w * G::NP + p0 + k]) : 1.f;
							v[k] = line_value_v(lb, (uint32_t)(jbase + k), m);
						}
					}
				} else {
					line_advance_hold(ls, len, lat, done);
				}
				team_sync<V>();
				if (!lazy) {
#pragma unroll
					for (int k = 0; k < T; ++k)
						if (owned[k]) slot_put<W, T>(out, w, p0 + k, v[k]);
				}
				if (tid == 0) {
					op->line[st.which] = ls;
					if (st.which == L_FREQ) { op->rt_fconst_valid = lazy ? 1u : 0u; op->rt_fconst = fc; }
					if (st.flags & SF_SKIP2) {
						LineState l2 = op->line[st.tmp];
						line_skip(l2, len, lat, done);
						op->line[st.tmp] = l2;
					}
				}
				break;
			}
			case ST_SMLINE: { /* generator.c:485-490 */
				float *out = slots + (size_t)st.out * G::SLOT;
				LineState ls = uni(op->line[L_PMA]);
				const bool active = (ls.v0 != 0.f) || (ls.flags & LP_GOAL);
				float v[T];
				if (active) {
					LineBlock lb = line_block_v(ls, len, false, 0.f);
					line_begin_state(ls, len, false, 0.f, lat, done);
#pragma unroll
					for (int k = 0; k < T; ++k)
						v[k] = owned[k] ? line_value_v(lb, (uint32_t)(jbase + k), 1.f) : 0.f;
				} else {
					line_skip(ls, len, lat, done);
#pragma unroll
					for (int k = 0; k < T; ++k) v[k] = 0.f;
				}
				team_sync<V>();
#pragma unroll
				for (int k = 0; k < T; ++k)
					if (owned[k]) slot_put<W, T>(out, w, p0 + k, v[k]);
				if (tid == 0) op->line[L_PMA] = ls;
				break;
			}
			case ST_LERP: { /* generator.c:466-467 */
				float *par = slots + (size_t)st.out * G::SLOT;
				const float *rpar = slots + (size_t)st.freq * G::SLOT;
				const float *mod = slots + (size_t)st.pm * G::SLOT;
				float v[T];
#pragma unroll
				for (int k = 0; k < T; ++k) {
					if (owned[k]) {
						int e = w * G::NP + p0 + k;
						float pv = par[e];
						pv += (rpar[e] - pv) * mod[e];
						v[k] = pv;
					}
				}
				team_sync<V>();
#pragma unroll
				for (int k = 0; k < T; ++k)
					if (owned[k]) slot_put<W, T>(par, w, p0 + k, v[k]);
				break;
			}
			case ST_OSC: {
				float *out = slots + (size_t)st.out * G::SLOT;
				float *scratch = slots; /* SCRATCH_SLOT */
				u32_alias *scratch_u = (u32_alias *)slots;
				const float *fslot = st.freq != NO_SLOT ? slots + (size_t)st.freq * G::SLOT : nullptr;
				const float *fmul = st.fmul != NO_SLOT ? slots + (size_t)st.fmul * G::SLOT : nullptr;
				const float *pmS = st.pm != NO_SLOT ? slots + (size_t)st.pm * G::SLOT : nullptr;
				const float *fpmS = st.fpm != NO_SLOT ? slots + (size_t)st.fpm * G::SLOT : nullptr;
				const float *ampS = st.amp != NO_SLOT ? slots + (size_t)st.amp * G::SLOT : nullptr;
				const float *smS = st.sm != NO_SLOT ? slots + (size_t)st.sm * G::SLOT : nullptr;
				const uint32_t type = uni(op->type);
				const bool is_osc = (type == OT_WAVE || type == OT_RASEG);
				const bool wave_env = (st.flags & SF_WAVE_ENV) != 0;
				const bool layer = (st.flags & SF_LAYER) != 0;

				float s[T], av[T], dv[T];
#pragma unroll
				for (int k = 0; k < T; ++k) { s[k] = 0.f; av[k] = 0.f; dv[k] = 0.f; }

				/* ---- frequency: one value for the block, a slot, or a line ---- */
				bool pconst = false; float pf = 0.f;
				if (st.prov != NO_SLOT) { pconst = uni(ops[st.prov].rt_fconst_valid) != 0; pf = uni(ops[st.prov].rt_fconst); }
				LineState fls, als, pls;
				LineBlock flb;
				const bool f_inline = is_osc && !fslot;
				bool fconst = false; float fc = 0.f;
				bool mconst = false; /* the ratio multiplier is a single value */
				if (is_osc) {
					if (f_inline) {
						fls = uni(op->line[L_FREQ]);
						fconst = const_freq(fls, fmul != nullptr, pconst, pf, fc);
						mconst = fmul && pconst;
						if (!fconst) {
							const float m0 = fmul ? (mconst ? pf : fmul[1]) : 0.f;
							flb = line_block_v(fls, len, fmul != nullptr, m0);
							line_begin_state(fls, len, fmul != nullptr, m0, lat, done);
						} else {
							line_advance_hold(fls, len, lat, done);
						}
					} else if (pconst) { /* own frequency block was never stored */
						fconst = true; fc = pf;
					}
				}
				/* ---- amplitude values, existing output for layering ---------- */
				const bool a_inline = !ampS;
				if (a_inline) {
					als = uni(op->line[L_AMP]);
					if (!(als.flags & LP_GOAL)) {
						const float ac = als.v0; /* held: sau/line.c:435-442 */
						line_advance_hold(als, len, lat, done);
#pragma unroll
						for (int k = 0; k < T; ++k) av[k] = ac;
					} else {
						const LineBlock alb = line_block_v(als, len, false, 0.f);
						line_begin_state(als, len, false, 0.f, lat, done);
#pragma unroll
						for (int k = 0; k < T; ++k)
							av[k] = owned[k] ? line_value_v(alb, (uint32_t)(jbase + k), 1.f) : 0.f;
					}
				} else {
#pragma unroll
					for (int k = 0; k < T; ++k)
						if (owned[k]) av[k] = ampS[w * G::NP + p0 + k];
				}
				if (layer) {
#pragma unroll
					for (int k = 0; k < T; ++k)
						if (owned[k]) dv[k] = out[w * G::NP + p0 + k];
				}
				bool sm_inline_active = false;
				LineState pls0; /* pm_a line before this block (the serial path re-derives values) */
				if (is_osc && (st.flags & SF_SM_INLINE)) {
					pls = uni(op->line[L_PMA]);
					pls0 = pls;
					sm_inline_active = (pls.v0 != 0.f) || (pls.flags & LP_GOAL);
					if (sm_inline_active) line_begin_state(pls, len, false, 0.f, lat, done);
					else line_skip(pls, len, lat, done);
				}
				const bool selfmod = is_osc && (smS != nullptr || sm_inline_active);

				/* state updates decided before barrier A, applied after it */
				uint32_t wb_phase = 0, wb_prev_phase = 0; double wb_prev_Is = 0; float wb_prev_s = 0;
				bool wb_owner = false;            /* this lane holds the block's last sample */
				bool wb_serial_done = false;      /* serial path already updated the osc state */
				unsigned long long wb_grand64 = 0;
				uint32_t wb_grand32 = 0;
				uint32_t wb_noise_prev = 0; bool wb_noise_prev_set = false;
				bool w_parallel = false;
				TabRef tab; tab.c23 = nullptr; tab.c01 = nullptr; tab.in_lds = false;
				WaveConst wc; wc.diff_scale = 0; wc.diff_offset = 0; wc.phase_adj = 0; wc.pad = 0;
				uint32_t ph[T];
#pragma unroll
				for (int k = 0; k < T; ++k) ph[k] = 0;

				if (type == OT_WAVE) {
					const uint32_t wave = uni(op->wave);
					wc.diff_scale = uni(misc->wc[wave].diff_scale);
					wc.diff_offset = uni(misc->wc[wave].diff_offset);
					{
						int ti = uni(misc->tab_of_wave[wave]);
						tab.in_lds = ti >= 0;
						tab.c23 = ti >= 0 ? t23 + (size_t)ti * WAVE_LEN : P.g_c23 + (size_t)wave * WAVE_LEN;
						tab.c01 = ti >= 0 ? t01 + (size_t)ti * WAVE_LEN : P.g_c01 + (size_t)wave * WAVE_LEN;
					}
					const float coeff = uni(op->coeff);
					const bool halo = (p0 == 0);
					const bool halo_live = halo && w > 0 && (jbase < (int)len);
					const uint32_t phase0 = uni(op->phase);
					uint32_t acc_last = 0; /* accumulator after this lane's last owned sample */
					if (fconst) {
						/* ---- wosc.h:135-169 with a constant increment: the wrapping
						 * sum of j+1 equal increments is one multiplication ---------- */
						const uint32_t inc = rint32w(coeff * fc);
#pragma unroll
						for (int k = 0; k < T; ++k) {
							const bool need = owned[k] || (k == 0 && halo_live);
							if (need) {
								const int e = w * G::NP + p0 + k;
								const uint32_t acc = phase0 + inc * (uint32_t)(jbase + k + 1);
								uint32_t ofs = (uint32_t)pm_offset(pmS != nullptr, fpmS != nullptr,
										pmS ? pmS[e] : 0.f, fpmS ? fpmS[e] : 0.f, fc, 0x1p31f);
								ph[k] = acc + ofs;
								if (owned[k]) acc_last = acc;
							}
						}
						wb_grand32 = inc * len;
					} else {
						/* ---- wosc.h:135-169: exact integer prefix scan -------------- */
						uint32_t inc[T], ofs[T];
						uint32_t lane_sum = 0;
#pragma unroll
						for (int k = 0; k < T; ++k) {
							inc[k] = 0; ofs[k] = 0;
							const bool need = owned[k] || (k == 0 && halo_live);
							if (need) {
								const int e = w * G::NP + p0 + k;
								const int j = jbase + k;
								float f = fslot ? fslot[e]
								                : line_value_v(flb, (uint32_t)j, fmul ? (mconst ? pf : fmul[e]) : 1.f);
								if (owned[k]) inc[k] = rint32w(coeff * f);
								ofs[k] = (uint32_t)pm_offset(pmS != nullptr, fpmS != nullptr,
										pmS ? pmS[e] : 0.f, fpmS ? fpmS[e] : 0.f, f, 0x1p31f);
							}
							lane_sum += inc[k];
						}
						const uint32_t incl = wave_incl_scan(lane_sum, l);
						if (l == 63) misc->tot32[w] = incl;
						team_sync<V>();
						uint32_t base = phase0;
#pragma unroll
						for (int ww = 0; ww < W; ++ww) {
							uint32_t t = misc->tot32[ww];
							if (ww < w) base += t;
							wb_grand32 += t;
						}
						uint32_t run = base + (incl - lane_sum);
#pragma unroll
						for (int k = 0; k < T; ++k) {
							run += inc[k];
							ph[k] = run + ofs[k];
							if (owned[k]) acc_last = run;
						}
					}
					if (!selfmod) {
						/* ---- lookup + differentiate: wosc.h:238-266 ------- */
						w_parallel = true;
						const bool reset = (op_flags & OPF_OSC_RESET) && len > 0;
						double Is[T];
						{
							/* sample before the block: carried state, or the
							 * one-table-step restart of wosc.h:215-231 */
							const uint32_t first = T > 1 ? ph[T > 1 ? 1 : 0] : __shfl_down(ph[0], 1);
							if (halo && w == 0) ph[0] = reset ? first - SLEN : op->prev_phase;
						}
#pragma unroll
						for (int k = 0; k < T; ++k) {
							const bool need = owned[k] || (k == 0 && (halo_live || (halo && w == 0 && reset)));
							Is[k] = need ? herp_lookup(tab, ph[k]) : 0.0;
						}
						if (halo && w == 0 && !reset) Is[0] = op->prev_Is;
						uint32_t pph = __shfl_up(ph[T - 1], 1);
						double pIs = __shfl_up(Is[T - 1], 1);
						bool anyzero = false;
#pragma unroll
						for (int k = 0; k < T; ++k) {
							if (k > 0) { pph = ph[k - 1]; pIs = Is[k - 1]; }
							if (owned[k]) {
								int32_t d = (int32_t)(ph[k] - pph);
								if (d == 0) anyzero = true;
								else s[k] = wosc_diff(Is[k], pIs, d, wc.diff_scale, wc.diff_offset);
								if (jbase + k == (int)len - 1) {
									wb_owner = true;
									wb_phase = acc_last; wb_prev_phase = ph[k];
									wb_prev_Is = Is[k]; wb_prev_s = s[k];
								}
							}
						}
						if (__any(anyzero) && l == 0) misc->flag = 1;
					} else {
						/* ---- serial: feedback recurrence, wosc.h:273-310 ---------- */
						team_sync<V>(); /* scratch may still be read as a slot by a lagging wave */
						/* everything that does not depend on the feedback is laid out
						 * first, in parallel: base phases in the scratch slot, the
						 * self-modulation amounts in a slot (the output slot is free
						 * until the combine step: its old contents are in dv[]) */
						const float *pmaS = smS;
						if (!pmaS) {
							LineBlock plb;
							plb = line_block_v(pls0, len, false, 0.f);
#pragma unroll
							for (int k = 0; k < T; ++k)
								if (owned[k]) out[w * G::NP + p0 + k] = line_value_v(plb, (uint32_t)(jbase + k), 1.f);
							pmaS = out;
						}
#pragma unroll
						for (int k = 0; k < T; ++k)
							if (owned[k]) scratch_u[w * G::NP + p0 + k] = ph[k];
						team_sync<V>();
						if (tid == 0 && len > 0) {
							uint32_t prev_phase = op->prev_phase;
							double prev_Is = op->prev_Is;
							float prev_s = op->prev_s, fb_s = op->fb_s;
							if (op->flags & OPF_OSC_RESET) {
								uint32_t phase00 = scratch_u[entry_of<W, T>(0)];
								prev_Is = herp_lookup(tab, phase00 - SLEN);
								double Is0 = herp_lookup(tab, phase00);
								prev_s = wosc_diff(Is0, prev_Is, (int32_t)SLEN, wc.diff_scale, wc.diff_offset);
								prev_Is = Is0;
								prev_phase = phase00;
							}
							SelfmodState ss;
							ss.prev_phase = prev_phase; ss.prev_Is = prev_Is; ss.prev_s = prev_s; ss.fb_s = fb_s;
							constexpr int SPAN = W > 1 ? G::NP - 1 : 0;
							if (tab.in_lds) selfmod_serial<true, SPAN>(tab, ss, wc, pmaS, scratch_u, len);
							else selfmod_serial<false, SPAN>(tab, ss, wc, pmaS, scratch_u, len);
							prev_phase = ss.prev_phase; prev_Is = ss.prev_Is; prev_s = ss.prev_s; fb_s = ss.fb_s;
							op->prev_phase = prev_phase;
							op->prev_Is = prev_Is;
							op->prev_s = prev_s;
							op->fb_s = fb_s;
						}
						team_sync<V>();
#pragma unroll
						for (int k = 0; k < T; ++k)
							if (owned[k]) s[k] = scratch[w * G::NP + p0 + k];
						wb_serial_done = true;
					}
				} else if (type == OT_RASEG) {
					/* ---- rasg.h:165-222 cycle|phase counter (post-increment) */
					const bool rate2x = (op->flags & OPF_RATE2X) != 0;
					const float coeff = rate2x ? op->coeff * 2 : op->coeff;
					const float phase_scale = rate2x ? 0x1p31f * 2 : 0x1p31f;
					const RasParams rp = ras_params(op->ras_func, op->ras_flags, op->ras_level,
							op->ras_alpha, op->wave);
					unsigned long long inc[T], ofs[T], lane_sum = 0;
#pragma unroll
					for (int k = 0; k < T; ++k) {
						inc[k] = 0; ofs[k] = 0;
						if (owned[k]) {
							const int e = w * G::NP + p0 + k;
							const int j = jbase + k;
							float f = fconst ? fc : (fslot ? fslot[e]
							                : line_value_v(flb, (uint32_t)j, fmul ? (mconst ? pf : fmul[e]) : 1.f));
							inc[k] = (unsigned long long)rint64(coeff * f);
							ofs[k] = (unsigned long long)pm_offset(pmS != nullptr, fpmS != nullptr,
									pmS ? pmS[e] : 0.f, fpmS ? fpmS[e] : 0.f, f, phase_scale);
						}
						lane_sum += inc[k];
					}
					const unsigned long long incl = wave_incl_scan64(lane_sum, l);
					if (l == 63) misc->tot64[w] = incl;
					team_sync<V>();
					unsigned long long base = op->cycle_phase;
#pragma unroll
					for (int ww = 0; ww < W; ++ww) {
						unsigned long long t = misc->tot64[ww];
						if (ww < w) base += t;
						wb_grand64 += t;
					}
					unsigned long long run = base + (incl - lane_sum);
					uint32_t cyc[T];
					float phf[T];
#pragma unroll
					for (int k = 0; k < T; ++k) {
						unsigned long long cp = ofs[k] + run;
						run += inc[k];
						ras_split(cp, cyc[k], phf[k]);
					}
					if (!selfmod) {
#pragma unroll
						for (int k = 0; k < T; ++k)
							if (owned[k]) s[k] = ras_sample(rp, cyc[k], phf[k], true); /* rasg.h:692-743 */
					} else {
						/* rasg.h:242-280 per-sample form with feedback */
						u32_alias *tmp = (u32_alias *)(slots + (size_t)st.tmp * G::SLOT);
#pragma unroll
						for (int k = 0; k < T; ++k) {
							if (owned[k]) {
								scratch[w * G::NP + p0 + k] = phf[k];
								tmp[w * G::NP + p0 + k] = cyc[k];
							}
						}
						team_sync<V>();
						if (tid == 0 && len > 0) {
							LineBlock plb;
							if (sm_inline_active) plb = line_block_v(pls0, len, false, 0.f);
							float fb_s = op->fb_s, prev_s = op->prev_s;
							for (uint32_t j = 0; j < len; ++j) {
								const uint32_t e = entry_of<W, T>(j);
								float pma_v = smS ? smS[e] : line_value_v(plb, j, 1.f);
								float pm_a = fb_s * pma_v * 0.5f;
								float phase = scratch[e] + pm_a;
								int32_t cycle_adj = (int32_t)floorf(phase);
								uint32_t cycle = tmp[e] + (uint32_t)cycle_adj;
								phase -= (float)cycle_adj;
								float sv = ras_sample(rp, cycle, phase, false);
								scratch[e] = sv;
								fb_s = ((fb_s + prev_s) + sv) * 0.5f; /* the reference build's association (see the oracle) */
								prev_s = sv;
							}
							op->fb_s = fb_s;
							op->prev_s = prev_s;
						}
						team_sync<V>();
#pragma unroll
						for (int k = 0; k < T; ++k)
							if (owned[k]) s[k] = scratch[w * G::NP + p0 + k];
					}
				} else if (type == OT_NOISE) {
					/* ---- noise.h:41-185 ---------------------------------- */
					const uint32_t nz = op->wave;
					const uint32_t n0 = op->noise_n;
					if (nz == NZ_re) {
						uint32_t term[T], lane_sum = 0;
#pragma unroll
						for (int k = 0; k < T; ++k) {
							term[k] = owned[k] ? (uint32_t)(((int32_t)ranfast32(n0 + (uint32_t)(jbase + k))) >> 6) : 0u;
							lane_sum += term[k];
						}
						const uint32_t incl = wave_incl_scan(lane_sum, l);
						if (l == 63) misc->tot32[w] = incl;
						team_sync<V>();
						uint32_t base = op->noise_prev, grand = 0;
#pragma unroll
						for (int ww = 0; ww < W; ++ww) {
							uint32_t t = misc->tot32[ww];
							if (ww < w) base += t;
							grand += t;
						}
						uint32_t run = base + (incl - lane_sum);
#pragma unroll
						for (int k = 0; k < T; ++k) {
							run += term[k];
							if (owned[k]) s[k] = fscalei((uint32_t)foldhd32((int32_t)run), 0x1p-31f);
						}
						wb_noise_prev = op->noise_prev + grand; wb_noise_prev_set = true;
					} else if (nz == NZ_vi) {
#pragma unroll
						for (int k = 0; k < T; ++k) {
							if (owned[k]) {
								uint32_t j = (uint32_t)(jbase + k);
								uint32_t s1 = ranfast32(n0 + j);
								uint32_t s0 = j == 0 ? op->noise_prev : ranfast32(n0 + j - 1);
								s[k] = fscalei((s1 / 2) - (s0 / 2), 0x1p-31f);
							}
						}
						if (len > 0) { wb_noise_prev = ranfast32(n0 + len - 1); wb_noise_prev_set = true; }
					} else if (nz == NZ_bv) {
#pragma unroll
						for (int k = 0; k < T; ++k) {
							if (owned[k]) {
								uint32_t j = (uint32_t)(jbase + k);
								int32_t s1 = noise_bv_term(n0 + j);
								int32_t s0 = j == 0 ? (int32_t)op->noise_prev : noise_bv_term(n0 + j - 1);
								s[k] = (float)(s1 - s0);
							}
						}
						if (len > 0) { wb_noise_prev = (uint32_t)noise_bv_term(n0 + len - 1); wb_noise_prev_set = true; }
					} else {
#pragma unroll
						for (int k = 0; k < T; ++k)
							if (owned[k]) s[k] = noise_stateless(nz, n0 + (uint32_t)(jbase + k));
					}
				} else { /* OT_AMP: generator.c:517-518 */
#pragma unroll
					for (int k = 0; k < T; ++k) s[k] = 1.f;
				}

				/* ---- barrier A: every read of operator state and input slots is done */
				team_sync<V>();
				if (w_parallel && uni(misc->flag) != 0) {
					/* rare: dphase == 0 somewhere -> hold the previous output
					 * (wosc.h:251-252), resolved serially over the block */
					team_sync<V>(); /* all flag reads done before it is cleared below */
#pragma unroll
					for (int k = 0; k < T; ++k)
						if (owned[k]) scratch_u[w * G::NP + p0 + k] = ph[k];
					team_sync<V>();
					if (tid == 0 && len > 0) {
						uint32_t prev_phase = op->prev_phase;
						double prev_Is = op->prev_Is;
						float prev_s = op->prev_s;
						if (op->flags & OPF_OSC_RESET) {
							uint32_t phase00 = scratch_u[entry_of<W, T>(0)];
							prev_Is = herp_lookup(tab, phase00 - SLEN);
							double Is0 = herp_lookup(tab, phase00);
							prev_s = wosc_diff(Is0, prev_Is, (int32_t)SLEN, wc.diff_scale, wc.diff_offset);
							prev_Is = Is0;
							prev_phase = phase00;
						}
						for (uint32_t j = 0; j < len; ++j) {
							const uint32_t e = entry_of<W, T>(j);
							uint32_t phase = scratch_u[e];
							int32_t d = (int32_t)(phase - prev_phase);
							float sv;
							if (d == 0) {
								sv = prev_s;
							} else {
								double Isv = herp_lookup(tab, phase);
								sv = wosc_diff(Isv, prev_Is, d, wc.diff_scale, wc.diff_offset);
								prev_Is = Isv; prev_s = sv; prev_phase = phase;
							}
							scratch[e] = sv;
						}
						op->prev_phase = prev_phase;
						op->prev_Is = prev_Is;
						op->prev_s = prev_s;
						misc->flag = 0;
					}
					team_sync<V>();
#pragma unroll
					for (int k = 0; k < T; ++k)
						if (owned[k]) s[k] = scratch[w * G::NP + p0 + k];
					wb_serial_done = true;
					wb_owner = false;
				}

				/* ---- combine (generator.c:384-440), hand-over, write-backs ------ */
				const bool to_voice = (st.which & OX_VOICE) != 0;
				LineState pl;
				LineBlock plb2;
				bool pan_goal = false;
				if (to_voice) { /* generator.c:749-788; the sum over voices is mix_kernel */
					pl = uni(op->line[L_PAN]);
					pan_goal = (pl.flags & LP_GOAL) != 0;
					if (pan_goal) { plb2 = line_block_v(pl, len, false, 0.f); line_begin_state(pl, len, false, 0.f, lat, done); }
					else line_skip(pl, len, lat, done);
				}
#pragma unroll
				for (int k = 0; k < T; ++k) {
					if (owned[k]) {
						const float r = mix_combine(dv[k], s[k], av[k], wave_env, layer);
						if (to_voice) {
							const int j = jbase + k;
							vrow[done + j] = r;
							if (prow) prow[done + j] = pan_goal ? line_value_v(plb2, (uint32_t)j, 1.f) : pl.v0;
						} else {
							slot_put<W, T>(out, w, p0 + k, r);
						}
					}
				}
				if (wb_owner) { /* the lane that holds the block's last sample */
					op->prev_phase = wb_prev_phase;
					op->prev_Is = wb_prev_Is;
					op->prev_s = wb_prev_s;
				}
				if (tid == 0) {
					if (type == OT_WAVE) {
						op->phase += wb_grand32;
						if (len > 0) op->flags &= ~OPF_OSC_RESET;
					}
					if (type == OT_RASEG) op->cycle_phase += wb_grand64;
					if (type == OT_NOISE) {
						op->noise_n += len;
						if (wb_noise_prev_set) op->noise_prev = wb_noise_prev;
					}
					if (f_inline) {
						op->line[L_FREQ] = fls;
						LineState l2 = op->line[L_FREQ2];
						line_skip(l2, len, lat, done);
						op->line[L_FREQ2] = l2;
						op->rt_fconst_valid = fconst ? 1u : 0u;
						op->rt_fconst = fc;
					}
					if (a_inline) {
						op->line[L_AMP] = als;
						LineState l2 = op->line[L_AMP2];
						line_skip(l2, len, lat, done);
						op->line[L_AMP2] = l2;
					}
					if (is_osc && (st.flags & SF_SM_INLINE)) op->line[L_PMA] = pls;
					if (to_voice) op->line[L_PAN] = pl;
				}
				if (to_voice) produced += len;
				(void)wb_phase; (void)wb_serial_done;
				break;
			}
			case ST_VOICE: { /* generator.c:749-788 with pan modulators */
				const float *src = slots + (size_t)st.out * G::SLOT;
				const float *panS = st.pm != NO_SLOT ? slots + (size_t)st.pm * G::SLOT : nullptr;
				LineState pl = op->line[L_PAN];
				LineBlock plb2;
				const bool pan_goal = !panS && (pl.flags & LP_GOAL);
				if (!panS) {
					if (pan_goal) { plb2 = line_block_v(pl, len, false, 0.f); line_begin_state(pl, len, false, 0.f, lat, done); }
					else line_skip(pl, len, lat, done);
				}
#pragma unroll
				for (int k = 0; k < T; ++k) {
					if (owned[k]) {
						const int j = jbase + k;
						const int e = w * G::NP + p0 + k;
						vrow[done + j] = src[e];
						if (prow)
							prow[done + j] = panS ? panS[e]
								: (pan_goal ? line_value_v(plb2, (uint32_t)j, 1.f) : pl.v0);
					}
				}
				team_sync<V>();
				if (tid == 0 && !panS) op->line[L_PAN] = pl;
				produced += len;
				break;
			}
			default:
				team_sync<V>();
				break;
			}

			if (st.flags & SF_END) { /* generator.c:719-728; runs after barrier A of ST_OSC */
				const bool inf = (op_flags & OPF_TIME_INF) != 0;
				--depth;
				const uint32_t outer = (st.flags & SF_BEGIN) ? parent_len : uni((uint32_t)misc->len_stack[depth]);
				if (!inf && !(st.flags & SF_LAYER) && !(st.which & OX_VOICE)) {
					float *out = slots + (size_t)st.out * G::SLOT;
#pragma unroll
					for (int k = 0; k < T; ++k) {
						int j = jbase + k;
						if (p0 + k >= 1 && j >= (int)len && j < (int)outer)
							slot_put<W, T>(out, w, p0 + k, 0.f);
					}
				}
				if (depth == 0 && st.op == vd.carr_local) {
					/* carrier finished: voice-level steps run for its length */
					cur_len = len;
					if (len == 0) block_ended = true; /* generator.c:842 */
				} else {
					/* (a pan modulator ending at the voice's level gives the length back to
					 * the steps after it: generator.c:762-771 run them for the carrier's) */
					cur_len = outer;
				}
				if (tid == 0 && !inf) op->time -= len;
			}
			/* ---- barrier B: stores and write-backs visible to the next step ---- */
			team_sync<V>();
		}
		done += blen;
	}

	if (tid == 0) { /* what the mixer needs to know about this row */
		VoiceOut vo;
		vo.pan_const = ops[vd.carr_local].line[L_PAN].v0;
		vo.has_pan = prow ? 1u : 0u;
		vo.valid_len = produced;
		vo.pan_row = vd.pan_dynamic_row;
		P.vinfo[vd.out_row] = vo;
	}
	team_sync<V>();
	for (uint32_t i = tid; i < vd.nops * 64; i += 64 * W)
		((u32_alias *)&P.ops[my_ids[i >> 6]])[i & 63] = ((const u32_alias *)ops)[i];
	} /* work list */
}


/* ======================================================================== */
/* time-parallel path: analyze -> fast -> finalize                          */
/* ======================================================================== */
/* While every line of a voice is held (no sweep pending), every oscillator
 * frequency is one value, nothing feeds back and no operator runs out of
 * time, sample t of the segment depends on the segment-start state only
 * through closed forms: phase(t) = phase0 + inc*(t+1) (the wrapping sum of
 * equal increments, wosc.h:129,145), noise counter n0 + t (noise.h:45).  Waves
 * then take chunks of the time axis independently: no barriers, no carried
 * state, block buffers private to the wave.  Each chunk recomputes H =
 * nesting-depth samples of lead-in so that the differentiators
 * (wosc.h:250-256) have their previous sample.  Everything else (sweeps, FM,
 * feedback, operators that expire) is left to render_kernel's block loop,
 * which continues where this path stops (fast_done). */

struct FastInfo {
	uint32_t total; /* frames this path renders (0: not eligible) */
	uint32_t H;     /* lead-in samples per chunk */
	uint32_t bail;  /* set when a chunk met dphase == 0 (hold-previous run) */
	uint32_t n_fsteps; /* decoded steps of the voice (decode_kernel): step list 0, the only or final pass */
	uint32_t n_pass[4]; /* ... of step lists 1..3 (sum passes) and 4 (chain-input pass) */
	uint32_t seq;   /* some oscillator's frequency varies (ramp, FM): phases are running sums. 1: one wave walks the
	                 * voice in order, carrying them; 2: two passes, every wave (no sum depends on another) */
	uint32_t n_scan; /* oscillators with running-sum phases (multi-pass voices) */
	uint32_t levels; /* deepest level among them (1: no sum depends on another) */
	uint32_t lvl_bits; /* 2 bits per such oscillator, in plan order: its level */
	uint32_t xlead;  /* lead-in lanes beyond the nesting depth (ratio frequencies below modulated blocks); in H */
	uint32_t n_chain; /* self-modulated oscillators handed to chain_kernel this segment */
};

struct FastStep;
struct FastLine;
struct FastAux;
struct ChainDesc;
constexpr uint32_t CHAIN_DESC_WORDS = 32;
constexpr uint32_t FAST_MAX_SCAN = 8;   /* oscillators with running-sum phases per multi-pass voice */
constexpr uint32_t LOOK_LDS_BYTES = FAST_MAX_SCAN * 2 * 64 * sizeof(unsigned long long); /* the look-back rings of a workgroup */
constexpr uint32_t FAST_MAX_LEVELS = 3; /* running sums that depend on running sums: at most that many sum passes
                                         * (FastParams.sum_levels of them are launched for a segment) */
/* A repeated phase (the output holds, wosc.h:251-252) on the first lane an operator's values are
 * defined in cannot take the held output from the lane before. What that spoils is exactly the
 * first owned frame of the row (one lane per nesting level upwards). fast_kernel notes such row
 * groups per voice and repair_kernel evaluates them once more FAST_REPAIR_SHIFT frames earlier,
 * where that frame lies in the middle of a row, storing only that frame. Through-zero PM makes
 * exact repeats a several-per-10-s event for a 1024-voice bank and one in 64 of them falls on
 * such a lane; each used to send its voice's whole segment to the block loop (8.5 ms for 10 s). */
constexpr uint32_t FAST_REPAIR_SHIFT = 24; /* H + shift < 64 (H <= 32) */
constexpr uint32_t FAST_MAX_REPAIR = 15;   /* noted row groups per voice and segment; more: block loop */
constexpr uint32_t FAST_REPAIR_WORDS = 2 + 2 * FAST_MAX_REPAIR; /* count, pad, then (group, rows) pairs */
constexpr uint32_t FAST_FLAGS = FAST_MAX_LEVELS + 3; /* pass_flags words */
/* Decoded steps are kept once per pass that runs them ([list][voice][step]): a pass walks its own list and never
 * loads a step only to find that another pass needs it (the per-step cost of the interpreter is most of a pass). */
constexpr uint32_t FAST_LISTS = 5; /* 0: only / final pass, 1..3: sum passes, 4: chain-input pass */
__device__ __forceinline__ uint32_t fast_list_of(uint32_t mode, uint32_t sum_levels) {
	return (mode == 0 || mode == sum_levels + 1) ? 0u : (mode == sum_levels + 2 ? 4u : mode);
}
constexpr uint32_t FR_CHAIN_IN = 4u << FAST_MAX_LEVELS; /* FastStep.ramp: the chain-input pass runs this step */
constexpr uint32_t FR_FINAL_SKIP = 8u << FAST_MAX_LEVELS; /* ... the final pass does not: only chains' inputs needed it */
constexpr uint32_t FT_CHAIN = 1u << 18;     /* FastStep.type: a feedback chain (rows = bits of FastStep.pan) */
constexpr uint32_t CHAIN_MARK = 0xC4A10001u; /* DevOp.ras_level of a W operator: chain_kernel staged its state */
struct FastParams {
	const VoiceDesc *voices;
	const Step *steps;
	const FastIds *fast_ids; /* parallel to steps */
	const uint32_t *op_ids;
	DevOp *ops;
	float *vout;
	float *pan;
	FastInfo *info;
	uint32_t *fast_done;
	uint32_t *worklist;   /* out: voices the block loop still has to run */
	uint32_t *work_count;
	VoiceOut *vinfo;
	const HerpC23 *g_c23;
	const HerpC01 *g_c01;
	FastStep *fsteps;     /* [n_voices][max_steps], written by decode_kernel */
	FastLine *flines;     /* same indexing: the ramp of a step whose line is in progress */
	FastAux *faux;        /* same indexing: sequential-scan extras */
	uint32_t row_stride, n_voices, n_fast, max_ops, max_steps, n_tabs, np;
	uint32_t rows;        /* T of the fast_kernel<T> that will run: block buffers hold 64 * rows frames */
	uint32_t enable;      /* 0: leave every voice to the block loop */
	uint32_t seq_enable;  /* block buffers are sized for frequency blocks: sequential-scan voices allowed */
	uint32_t ids_full_ofs;/* offset of the with-frequency numbering in fast_ids */
	uint32_t mode;        /* fast_kernel: 0 the only pass; 1..sum_levels: sums of phase increments of that level;
	                       * sum_levels + 1: final pass. scan_kernel: the level whose sums to prefix */
	uint32_t sum_levels;  /* sum passes this segment's launch sequence has (2, or 3 when the host expects that depth) */
	unsigned long long *scan; /* [n_voices][FAST_MAX_SCAN][scan_groups]: those sums (W: mod 2^32; R: 64 bits), then
	                           * (scan_kernel) their prefixes */
	uint32_t scan_groups;
	uint32_t *pass_flags; /* [FAST_MAX_LEVELS]: some voice of the segment needs that sum pass (set by analyze_kernel);
	                       * [FAST_MAX_LEVELS]: some voice has row groups noted for repair_kernel;
	                       * [FAST_MAX_LEVELS + 1]: some voice has feedback chains */
	uint32_t *repair;     /* [voice][FAST_REPAIR_WORDS] */
	uint32_t repair_on;   /* 0: such voices go to the block loop (SAU_AMD_NO_REPAIR, tests) */
	/* feedback recurrences (wosc.h:273-310) out of the time-parallel passes: a pair of rows per chain in HBM --
	 * base phases, then (in place) the samples; self-modulation amounts -- and what chain_kernel needs to run it */
	float *chain_rows;    /* [n_chain_rows][2][chain_stride], or NULL: such voices go to the block loop */
	uint32_t chain_stride, n_chain_rows;
	ChainDesc *chain_desc;
	FastLine *fplines;    /* [voice][max_steps]: the self-modulation amount line of a chain step without a block for it */
	uint32_t n_ctabs;     /* wave tables chain_kernel stages in LDS */
	uint32_t chain_inline;/* chains fed from their own lines by chain_kernel's feeder wave (SAU_AMD_CHAIN_INLINE; off:
	                       * measured slower) */
	/* A segment with chains is pipelined in chunks of frames: while chain_kernel (64 CUs, a second stream) runs
	 * chunk c, the chain-input pass prepares chunk c + 1 and the final pass finishes chunk c - 1 on the other CUs.
	 * fast_kernel: range_mode 1 = the row groups that start in [f_lo, f_hi), 2 = those that end in (f_lo, f_hi]
	 * (0: all). chain_kernel: frames [f_lo, f_hi) of every chain, continuing from the staged state when f_lo > 0. */
	uint32_t range_mode, f_lo, f_hi, range_last;
	/* Saved phase increments: a running-sum oscillator's per-frame increments, computed in the sum pass of its
	 * level, go to a row pair in HBM (W: 32 bits in the first row; R: low and high words), and the final pass
	 * reads them back instead of evaluating the frequency again -- whatever only produced that frequency (FM
	 * modulators, their sub-trees) is then left out of the final pass. */
	uint32_t *inc_rows;   /* [n_inc_rows][2][inc_stride], or NULL */
	uint32_t inc_stride, n_inc_rows;
	/* Single-pass running sums (seq kind 3): one word per oscillator and row group (R: two, low and high half),
	 * {epoch:30, status:2, value:32}; a wave publishes its group's sum (status 1), adds up what its predecessors
	 * have published back to the nearest finished prefix, and publishes its own prefix (status 2). The epoch
	 * (one per segment) makes every older word read as empty, so nothing is cleared between segments. */
	unsigned long long *look; /* [n_look_rows][2][scan_groups] (VoiceDesc.look_base/n_look), or NULL */
	uint32_t look_epoch;
	uint32_t rows_multi; /* rows per pass in the launches of the full running-sum build (kinds 1 and 2) */
	uint32_t only_multi; /* this launch: only the voices fast_kernel<T, 2> leaves out (one wave in order, several passes) */
	int8_t ctab_of_wave[12];
	uint8_t cwave_of_tab[12];
	int8_t tab_of_wave[12];
	uint8_t wave_of_tab[12];
	WaveConst wc[12];
};

/* a W oscillator step whose self-modulation is on (generator.c:479-498, wosc.h:273-310): chain_kernel's */
__device__ __forceinline__ bool step_is_chain(const Step &st, const DevOp &o) {
	return !o.rt_frozen && step_may_chain(st) && o.type == OT_WAVE &&
		(st.sm != NO_SLOT || o.line[L_PMA].v0 != 0.f || (o.line[L_PMA].flags & LP_GOAL));
}
/* ... and whose varying frequency is its only phase input: chain_kernel sums the phase increments itself
 * (a sum pass and a scan less), the chain-input pass hands it increments instead of base phases */
__device__ __forceinline__ bool step_is_chain_acc(const Step &st, const DevOp &o) {
	return step_is_chain(st, o) && !o.rt_fconst_valid && st.pm == NO_SLOT && st.fpm == NO_SLOT;
}

/* ... and all of whose inputs are its own lines: a frequency that is one value, or its frequency line alone
 * (times a parent frequency that is one value) with nothing added into its block, amounts from its pm_a line.
 * chain_kernel's feeder wave evaluates those itself; the chain-input pass has nothing to do for it.
 * line_step: the plan index of the ST_LINE step that fills its frequency block, or ~0u. */
__device__ __forceinline__ bool step_is_chain_inline(bool enabled, const Step *plan, uint32_t si, const uint32_t *ids, const DevOp *ops,
		uint32_t *line_step) {
	if (!enabled) { *line_step = ~0u; return false; }
	const Step st = plan[si];
	const DevOp &o = ops[ids[st.op]];
	*line_step = ~0u;
	if (!step_is_chain(st, o) || st.pm != NO_SLOT || st.fpm != NO_SLOT || st.sm != NO_SLOT) return false;
	if (o.rt_fconst_valid) return true;
	uint32_t fmul = st.fmul, prov = st.prov;
	if (st.freq != NO_SLOT) {
		uint32_t q = si;
		bool found = false;
		while (q-- > 0) { /* its block: written by its own line step and by nothing since */
			const Step sq = plan[q];
			if (sq.kind == ST_LINE && sq.which == L_FREQ && sq.op == st.op && sq.out == st.freq) { found = true; break; }
			if ((sq.kind == ST_OSC || sq.kind == ST_LERP || sq.kind == ST_LINE || sq.kind == ST_SMLINE) && sq.out == st.freq) return false;
		}
		if (!found) return false;
		*line_step = q;
		fmul = plan[q].fmul; prov = plan[q].prov;
	}
	if (fmul != NO_SLOT) { /* a ratio of the parent's frequency: only when that is one value */
		const LineState &fl = o.line[L_FREQ];
		const bool ratio = (fl.flags & LP_STATE_RATIO) || ((fl.flags & LP_GOAL) && (fl.flags & LP_GOAL_RATIO));
		if (ratio && !(prov != NO_SLOT && ops[ids[prov]].rt_fconst_valid)) return false;
	}
	return true;
}

/* the operator whose frequency line most recently filled block `slot` before step si (0xff: none) */
__device__ __forceinline__ uint32_t block_owner(const Step *plan, uint32_t si, uint32_t slot) {
	for (uint32_t q = si; q-- > 0;) {
		const Step sq = plan[q];
		if (sq.kind == ST_LINE && sq.which == L_FREQ && sq.out == slot) return sq.op;
	}
	return 0xff;
}

/* does voice-local operator `op` take frequency-scaled phase modulation? */
__device__ __forceinline__ bool op_has_fpm(const Step *plan, uint32_t n, uint32_t op) {
	for (uint32_t q = 0; q < n; ++q) {
		const Step sq = plan[q];
		if (sq.kind == ST_OSC && sq.op == op) return sq.fpm != NO_SLOT;
	}
	return false;
}

__global__ void __launch_bounds__(64) analyze_kernel(FastParams P) {
	const uint32_t v = blockIdx.x * 64 + threadIdx.x;
	if (v == 0) *P.work_count = 0; /* finalize_kernel (a later launch) builds the block loop's work list */
	if (v >= P.n_voices) return;
	const VoiceDesc vd = P.voices[v];
	const uint32_t *ids = P.op_ids + vd.ops_ofs;
	bool bad = (vd.flags & VD_NO_FAST) != 0 || !P.enable;
	bool seq = false;
	uint32_t min_time = 0xFFFFFFFFu;
	const Step *plan = P.steps + vd.plan_ofs;
	/* An operator that has run out of time yields nothing, and neither it nor
	 * anything nested in it advances (run_block gives its subtree zero
	 * length, generator.c:686-700): such subtrees are left out below. */
	for (uint32_t i = 0; i < vd.nops; ++i) P.ops[ids[i]].rt_frozen = 0;
	if (P.chain_desc)
		for (uint32_t k = 0; k < vd.n_chain; ++k) /* ChainDesc.n (its first word; the type is defined further down) */
			((uint32_t *)P.chain_desc)[(size_t)(vd.chain_base + k) * CHAIN_DESC_WORDS] = 0;
	const bool chain_ok = P.chain_rows != nullptr && P.scan != nullptr;
	bool has_chain = false;
	{
		uint32_t dep = 0, frozen_at = 0;
		for (uint32_t si = 0; si < vd.plan_len; ++si) {
			const Step st = plan[si];
			DevOp &o = P.ops[ids[st.op]];
			if (st.flags & SF_BEGIN) {
				++dep;
				if (!frozen_at && !(o.flags & OPF_TIME_INF) && o.time == 0) frozen_at = dep;
			}
			if (frozen_at) o.rt_frozen = 1;
			if (st.flags & SF_END) {
				if (dep == frozen_at) frozen_at = 0;
				--dep;
			}
		}
	}
	if (P.ops[ids[vd.carr_local]].rt_frozen) bad = true; /* the voice is over (generator.c:839) */
	for (uint32_t i = 0; i < vd.nops; ++i) {
		DevOp &o = P.ops[ids[i]];
		if (o.rt_frozen) continue;
		/* ramps in progress: amplitude lines are closed-form per frame (sau/line.c
		 * fills depend on the position only); frequency ramps need a phase scan,
		 * self-modulation and pan ramps stay with the block loop */
		for (uint32_t ln = 0; ln < L_COUNT; ++ln) {
			if (!(o.line[ln].flags & LP_GOAL)) continue;
			if (ln == L_FREQ || ln == L_FREQ2) seq = true; /* phase becomes a running sum */
			else if (ln == L_PAN) { /* fine when the plan gives the pan line a step of its own */
				if (!(vd.plan_len && plan[vd.plan_len - 1].kind == ST_VOICE)) bad = true;
			} else if (ln == L_PMA) { if (!(chain_ok && o.type == OT_WAVE)) bad = true; }
			else if (ln != L_AMP && ln != L_AMP2) bad = true;
		}
		if (o.type == OT_NOISE && o.wave == NZ_re) bad = true;
		/* self-modulation is a recurrence: W oscillators' go to chain_kernel, R's to the block loop */
		if (o.line[L_PMA].v0 != 0.f && !(chain_ok && o.type == OT_WAVE)) bad = true;
		if (o.type == OT_WAVE) o.ras_level = 0; /* (CHAIN_MARK of an earlier segment) */
		o.rt_fconst_valid = 0;
		o.rt_fblk_valid = 0;
		o.st_phase = 0; /* until the kernels stage into it: see "modulated blocks" below */
		o.st_prev_phase = 0; /* likewise: extra lead-in of the operator while this kernel and decode_kernel run */
		if (!(o.flags & OPF_TIME_INF) && o.time < min_time) min_time = o.time;
	}
	uint32_t depth = 0, maxd = 0;
	/* Extra lead-in per block buffer, over what its writer's nesting depth gives: contents exact
	 * from lane H - depth + 1 + extra. 0..7 in three bit planes over the 256 buffer ids. It
	 * arises where a ratio frequency multiplies by a modulated frequency block written at a
	 * smaller depth than the reader's (see "modulated block" below) and travels up the
	 * operator tree with the data. */
	unsigned long long x0[4] = {0, 0, 0, 0}, x1[4] = {0, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
	auto extra_of = [&](uint32_t sl) -> uint32_t {
		if (sl == NO_SLOT) return 0;
		const uint32_t q = sl >> 6, sh = sl & 63;
		const unsigned long long a = q == 0 ? x0[0] : q == 1 ? x0[1] : q == 2 ? x0[2] : x0[3];
		const unsigned long long b = q == 0 ? x1[0] : q == 1 ? x1[1] : q == 2 ? x1[2] : x1[3];
		const unsigned long long c = q == 0 ? x2[0] : q == 1 ? x2[1] : q == 2 ? x2[2] : x2[3];
		return (uint32_t)((a >> sh) & 1ull) | ((uint32_t)((b >> sh) & 1ull) << 1) | ((uint32_t)((c >> sh) & 1ull) << 2);
	};
	auto set_extra = [&](uint32_t sl, uint32_t x, bool keep_max) {
		if (sl == NO_SLOT) return;
		if (keep_max) { const uint32_t old = extra_of(sl); if (old > x) x = old; }
		const unsigned long long bit = 1ull << (sl & 63);
#pragma unroll
		for (int q = 0; q < 4; ++q)
			if ((int)(sl >> 6) == q) {
				x0[q] = (x0[q] & ~bit) | ((x & 1) ? bit : 0ull);
				x1[q] = (x1[q] & ~bit) | ((x & 2) ? bit : 0ull);
				x2[q] = (x2[q] & ~bit) | ((x & 4) ? bit : 0ull);
			}
	};
	uint32_t x_carrier = 0;
	for (uint32_t si = 0; si < vd.plan_len && !bad; ++si) {
		const Step st = plan[si];
		DevOp &o = P.ops[ids[st.op]];
		if (o.rt_frozen) { /* nesting still counts: depths of live steps stay what they are */
			if (st.flags & SF_BEGIN) ++depth;
			if (st.flags & SF_END) --depth;
			continue;
		}
		if (st.flags & SF_BEGIN) { ++depth; if (depth > maxd) maxd = depth; }
		const bool is_osc = o.type == OT_WAVE || o.type == OT_RASEG;
		const bool freq_here = (st.kind == ST_LINE && st.which == L_FREQ) ||
			(st.kind == ST_OSC && st.freq == NO_SLOT && is_osc);
		if (st.kind == ST_ZERO) bad = true;
		if (st.kind == ST_SMLINE && !(chain_ok && o.type == OT_WAVE)) bad = true;
		if (st.kind == ST_OSC && st.sm != NO_SLOT && !(chain_ok && o.type == OT_WAVE)) bad = true;
		if (step_may_chain(st) && o.type == OT_WAVE &&
		    (st.sm != NO_SLOT || o.line[L_PMA].v0 != 0.f || (o.line[L_PMA].flags & LP_GOAL)))
			has_chain = true;
		/* a ratio line (sau/line.c:72) multiplies by the parent's frequency: one value, or a block */
		bool pconst = false; float pf = 0.f;
		if (st.fmul != NO_SLOT && st.fmul >= FSLOT_BASE) {
			/* the parent's frequency block as it stands when this step reads it: one value if
			 * the parent's line is held and nothing has been added into the block yet (the
			 * first FM modulator of a plain carrier sees exactly that, generator.c:448-477) */
			const uint32_t ow = block_owner(plan, si, st.fmul);
			if (ow != 0xff) {
				const DevOp &po = P.ops[ids[ow]];
				pconst = po.rt_fblk_valid != 0; pf = po.rt_fconst;
			} else if (st.prov != NO_SLOT) {
				const DevOp &po = P.ops[ids[st.prov]];
				pconst = po.rt_fconst_valid != 0; pf = po.rt_fconst;
			}
		}
		/* Anything added into a frequency block makes it per-frame: a modulated block. Its
		 * contents are exact only from the lane its writers are (nesting depth wd: lane
		 * H - wd + 1). Whoever multiplies by it -- ratio lines of operators nested in its
		 * owner, directly or through blocks derived from it -- must not need it earlier:
		 * a reader at depth d sums increments from lane H - d + 1 on (one earlier when it
		 * scales a phase modulator by its frequency). A deeper reader gets that many lanes of
		 * extra lead-in (and so does everything that consumes its output, up to the carrier:
		 * the voice's rows get H + extra lead-in lanes).
		 * (st_phase holds wd while this kernel runs; 0: not a modulated block.) */
		uint32_t x_step = 0; /* this step's output: extra lead-in of what it reads */
		if (st.kind == ST_LERP) x_step = extra_of(st.freq) > extra_of(st.pm) ? extra_of(st.freq) : extra_of(st.pm);
		if ((st.kind == ST_OSC || st.kind == ST_LERP) && st.out != NO_SLOT && st.out >= FSLOT_BASE) {
			const uint32_t ow = block_owner(plan, si, st.out);
			if (ow != 0xff) {
				DevOp &oo = P.ops[ids[ow]];
				oo.rt_fblk_valid = 0;
				const uint32_t wd = st.kind == ST_OSC ? depth : depth + 1;
				if (oo.st_phase == 0 || wd < oo.st_phase) oo.st_phase = wd;
			}
		}
		if (st.kind == ST_LINE || freq_here) {
			const LineState &ls = o.line[st.kind == ST_LINE ? st.which : L_FREQ];
			const bool g_ratio = (ls.flags & LP_GOAL_RATIO) != 0, s_ratio = (ls.flags & LP_STATE_RATIO) != 0;
			if (st.fmul != NO_SLOT) {
				/* a ramp whose goal and state disagree about being ratios rescales its
				 * state by the parent's first sample (sau/line.c:358-370): block loop */
				if ((ls.flags & LP_GOAL) && g_ratio != s_ratio) {
					/* fine when the parent's frequency is one value for the segment: decode_kernel and
					 * finalize_kernel then apply the rescaling with it */
					const bool parent_const = st.prov != NO_SLOT && P.ops[ids[st.prov]].rt_fconst_valid != 0;
					if (!parent_const) bad = true;
				}
				if ((s_ratio || ((ls.flags & LP_GOAL) && g_ratio)) && !pconst) {
					seq = true;
					if (st.fmul >= FSLOT_BASE) {
						const uint32_t ow = block_owner(plan, si, st.fmul);
						const uint32_t wd = ow != 0xff ? P.ops[ids[ow]].st_phase : 0u;
						if (wd) {
							const uint32_t need = depth + (op_has_fpm(plan, vd.plan_len, st.op) ? 1u : 0u);
							/* the block is exact from lane H - wd + 1 + its own extra; this reader
							 * would sum from lane H - need + 1 */
							x_step = (need > wd ? need - wd : 0u) + extra_of(st.fmul);
							/* this operator's own block derives from the modulated one */
							if (st.kind == ST_LINE && st.which == L_FREQ && (o.st_phase == 0 || wd < o.st_phase))
								o.st_phase = wd;
						}
					}
				}
			}
		}
		if (freq_here && !bad) {
			const LineState &fl = o.line[L_FREQ];
			/* one value for the segment? (the block loop's const_freq) */
			bool isconst = !(fl.flags & LP_GOAL) && !(st.kind == ST_LINE && (st.flags & SF_FORCE));
			float fc = fl.v0;
			if (st.fmul != NO_SLOT && (fl.flags & LP_STATE_RATIO)) {
				if (pconst) fc = fl.v0 * pf; /* sau/line.c:72 */
				else isconst = false;
			}
			o.rt_fconst = fc;
			o.rt_fconst_valid = isconst ? 1u : 0u;
			/* the block itself (before modulators are added) holds one value? */
			o.rt_fblk_valid = (isconst || (st.kind == ST_LINE && (st.flags & SF_FORCE) && !(fl.flags & LP_GOAL) &&
					!((fl.flags & LP_STATE_RATIO) && st.fmul != NO_SLOT && !pconst))) ? 1u : 0u;
			if (!isconst) seq = true;
		}
		if (st.kind == ST_OSC && is_osc && st.freq != NO_SLOT && !o.rt_fconst_valid) seq = true;
		/* extra lead-in: what this step reads, what its own ratio frequency needs, to what it writes */
		if (st.kind == ST_LINE) {
			/* the block made here is as exact as the one it multiplies by; what the operator itself
			 * needs on top waits in st_prev_phase for its oscillator step */
			if (x_step > 7) bad = true;
			set_extra(st.out, extra_of(st.fmul), false);
			if (st.which == L_FREQ) o.st_prev_phase = x_step;
		} else if (st.kind == ST_LERP) {
			if (x_step > 7) bad = true;
			set_extra(st.out, x_step, true);
		} else if (st.kind == ST_OSC) {
			uint32_t x = x_step > o.st_prev_phase ? x_step : o.st_prev_phase; /* its own frequency's need */
			const uint32_t in[5] = {extra_of(st.freq), extra_of(st.pm), extra_of(st.fpm), extra_of(st.amp), extra_of(st.sm)};
#pragma unroll
			for (int k = 0; k < 5; ++k) if (in[k] > x) x = in[k];
			if (x > 7) bad = true;
			o.st_prev_phase = x; /* for decode_kernel (the kernels stage into this field only later) */
			if (st.op == vd.carr_local) x_carrier = x;
			if (!(st.which & OX_VOICE)) set_extra(st.out, x, (st.flags & SF_LAYER) != 0 || st.out >= FSLOT_BASE);
		} else if (st.kind == ST_VOICE) {
			const uint32_t x = extra_of(st.out) > extra_of(st.pm) ? extra_of(st.out) : extra_of(st.pm);
			if (x > x_carrier) x_carrier = x;
		}
		if (st.flags & SF_END) --depth;
	}
	/* Two passes suffice when no running sum depends on another one: the per-frame
	 * increments of every such oscillator (its frequency inputs) must not depend on
	 * the output of an oscillator whose phase is itself a running sum. Forward
	 * data-flow over the block buffers ("tainted" = depends on such an output). */
	/* Several passes instead of one wave in order: a running sum can be computed by all
	 * waves once the sums it depends on are known. Level 1: its per-frame increments (its
	 * frequency inputs) depend on no other running-sum oscillator's output; level n + 1:
	 * they depend on level-n outputs. Forward data-flow over the block buffers, two bits
	 * per buffer: the deepest level its contents depend on. */
	/* Feedback chains: the recurrence's inputs (frequency, phase modulators, amounts) must not depend on any
	 * chain's output, and no running sum may either -- the sum passes and the chain-input pass run before
	 * chain_kernel. Forward data-flow, one bit per block buffer ("depends on a chain's output"). */
	if (has_chain && !bad) {
		unsigned long long c0[4] = {0, 0, 0, 0};
		auto dep = [&](uint32_t sl) -> bool {
			if (sl == NO_SLOT) return false;
			const uint32_t q = sl >> 6;
			const unsigned long long a = q == 0 ? c0[0] : q == 1 ? c0[1] : q == 2 ? c0[2] : c0[3];
			return ((a >> (sl & 63)) & 1ull) != 0;
		};
		auto set_dep = [&](uint32_t sl, bool v, bool keep) {
			if (sl == NO_SLOT) return;
			const unsigned long long bit = 1ull << (sl & 63);
#pragma unroll
			for (int q = 0; q < 4; ++q)
				if ((int)(sl >> 6) == q) c0[q] = v ? (c0[q] | bit) : (keep ? c0[q] : (c0[q] & ~bit));
		};
		for (uint32_t si = 0; si < vd.plan_len && !bad; ++si) {
			const Step st = plan[si];
			const DevOp &o = P.ops[ids[st.op]];
			if (o.rt_frozen) continue;
			if (st.kind == ST_LINE) set_dep(st.out, dep(st.fmul), false);
			else if (st.kind == ST_SMLINE) set_dep(st.out, false, false);
			else if (st.kind == ST_LERP) set_dep(st.out, dep(st.freq) || dep(st.pm), true);
			else if (st.kind == ST_OSC) {
				const bool in_dep = dep(st.pm) || dep(st.fpm) || dep(st.freq) || dep(st.fmul) || dep(st.sm);
				const bool chain = step_may_chain(st) && o.type == OT_WAVE &&
					(st.sm != NO_SLOT || o.line[L_PMA].v0 != 0.f || (o.line[L_PMA].flags & LP_GOAL));
				const bool fvar = (o.type == OT_WAVE || o.type == OT_RASEG) && !o.rt_fconst_valid;
				if (chain && in_dep) bad = true;
				if (fvar && (dep(st.freq) || dep(st.fmul))) bad = true;
				if (!(st.which & OX_VOICE)) set_dep(st.out, chain || in_dep || dep(st.amp), (st.flags & SF_LAYER) != 0);
			}
		}
	}
	uint32_t seq_kind = seq ? 1u : 0u, n_scan_out = 0, levels_out = 0, lvl_bits_out = 0;
	if (seq && !bad) {
		unsigned long long t0[4] = {0, 0, 0, 0}, t1[4] = {0, 0, 0, 0};
		auto level_of = [&](uint32_t sl) -> uint32_t {
			if (sl == NO_SLOT) return 0;
			const uint32_t q = sl >> 6;
			const unsigned long long a = q == 0 ? t0[0] : q == 1 ? t0[1] : q == 2 ? t0[2] : t0[3];
			const unsigned long long b = q == 0 ? t1[0] : q == 1 ? t1[1] : q == 2 ? t1[2] : t1[3];
			return (uint32_t)((a >> (sl & 63)) & 1ull) | ((uint32_t)((b >> (sl & 63)) & 1ull) << 1);
		};
		auto set_level = [&](uint32_t sl, uint32_t lv, bool keep_max) {
			if (sl == NO_SLOT) return;
			if (keep_max) { const uint32_t old = level_of(sl); if (old > lv) lv = old; }
			const unsigned long long bit = 1ull << (sl & 63);
#pragma unroll
			for (int q = 0; q < 4; ++q)
				if ((int)(sl >> 6) == q) {
					t0[q] = (t0[q] & ~bit) | ((lv & 1) ? bit : 0ull);
					t1[q] = (t1[q] & ~bit) | ((lv & 2) ? bit : 0ull);
				}
		};
		auto max2 = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
		bool multi = true;
		uint32_t n_scan = 0;
		for (uint32_t si = 0; si < vd.plan_len; ++si) {
			const Step st = plan[si];
			DevOp &o = P.ops[ids[st.op]];
			if (o.rt_frozen) continue;
			if (st.kind == ST_LINE) {
				set_level(st.out, level_of(st.fmul), false);
			} else if (st.kind == ST_LERP) {
				set_level(st.out, max2(level_of(st.freq), level_of(st.pm)), true);
			} else if (st.kind == ST_OSC) {
				const bool fvar = (o.type == OT_WAVE || o.type == OT_RASEG) && !o.rt_fconst_valid &&
					!(chain_ok && step_is_chain_acc(st, o));
				uint32_t lv = max2(max2(level_of(st.pm), level_of(st.fpm)), max2(level_of(st.amp),
						max2(level_of(st.freq), level_of(st.fmul))));
				if (fvar) {
					const uint32_t mine = 1 + max2(level_of(st.freq), level_of(st.fmul));
					if (mine > P.sum_levels || n_scan >= FAST_MAX_SCAN) multi = false; /* deeper: one wave, in order */
					else {
						lvl_bits_out |= mine << (2 * n_scan);
						if (mine > levels_out) levels_out = mine;
						o.rt_fblk_valid = mine; /* (its first meaning is over: from here on the operator's level) */
					}
					++n_scan;
					lv = max2(lv, mine > 3 ? 3u : mine);
				}
				if (!(st.which & OX_VOICE)) set_level(st.out, lv, (st.flags & SF_LAYER) != 0);
			}
		}
		if (P.look && !has_chain && n_scan <= FAST_MAX_SCAN && n_scan <= vd.n_look) {
			seq_kind = 3; /* one pass, any depth: prefixes by look-back */
		} else if (multi && P.scan) {
			seq_kind = 2;
			for (uint32_t p = 0; p < levels_out && p < P.sum_levels; ++p) atomicOr(&P.pass_flags[p], 1u);
		}
		n_scan_out = n_scan;
	}
	if (has_chain && !bad) {
		if (seq_kind == 1) bad = true; /* (one wave in order: not with chains) */
		else {
			seq_kind = 2; /* multi-pass voice, possibly without sums */
			atomicOr(&P.pass_flags[FAST_MAX_LEVELS + 1], 1u);
		}
	}
	FastInfo fi;
	/* Running-sum voices get one more lead-in lane than their data flow needs: a repeated phase on
	 * the first lane an operator is defined in then spoils nothing that is stored (what it spoils
	 * climbs one lane per nesting level and ends on the lane before the first stored one), where
	 * closed-form voices have repair_kernel for that case (see FAST_REPAIR_SHIFT). */
	if (seq || has_chain) ++x_carrier;
	fi.n_chain = has_chain && !bad ? 1u : 0u;
	fi.H = maxd + x_carrier; fi.xlead = x_carrier; fi.bail = 0; fi.n_fsteps = 0; fi.n_pass[0] = fi.n_pass[1] = fi.n_pass[2] = fi.n_pass[3] = 0; fi.seq = seq_kind; fi.n_scan = n_scan_out; fi.levels = levels_out; fi.lvl_bits = lvl_bits_out;
	fi.total = 0;
	if ((seq || has_chain) && !P.seq_enable) bad = true;
	if (!bad && vd.nops <= P.max_ops && vd.plan_len <= P.max_steps && maxd >= 1 && maxd + x_carrier <= P.np / 2)
		fi.total = min(min_time, vd.run_len);
	P.info[v] = fi;
	if (fi.total && (seq_kind == 1 || seq_kind == 2)) atomicOr(&P.pass_flags[FAST_MAX_LEVELS + 2], 1u);
	P.fast_done[v] = 0;
	P.repair[(size_t)v * FAST_REPAIR_WORDS] = 0;
}

/* One step of a voice's plan, decoded once per voice per wave into immediate
 * form (LDS offsets, constants) so that a chunk touches no operator records:
 * lazily-constant frequency lines vanish, everything a step needs is 20 dwords. */
struct FastStep {
	uint32_t kind;      /* ST_* | flags << 8 | which << 16 | depth << 24 */
	uint32_t out_off, pm_off, fpm_off, amp_off, aux_off; /* float offsets in the wave's slot area, ~0u = none */
	uint32_t type;      /* OT_* | wave/noise id << 8 | reset << 16 */
	uint32_t inc, phase0, prev_phase; /* N: noise_n in phase0, noise_prev in prev_phase */
	float fc, ac, diff_scale, diff_offset;
	int32_t tab;        /* index of the staged table, or -1 */
	uint32_t gop;       /* global operator index (state staging) */
	double prev_Is;
	float pan;
	uint32_t ramp;      /* bit 0: the step's line is a ramp in progress (FastLine); bit 1: FastAux present;
	                     * bit 2 + p: sum pass p + 1 of a multi-pass voice runs this step */
};
static_assert(sizeof(FastStep) == 80, "FastStep is 20 dwords");

/* A line block over the whole segment (line_begin): frames [0, goal_len)
 * follow the sweep, later ones hold. */
struct FastLine {
	Sweep sw;
	uint32_t goal_len;
	float hold;
	uint32_t pad;
};
static_assert(sizeof(FastLine) == 48, "FastLine is 12 dwords");

/* What chain_kernel needs to run one feedback chain for a segment (written by decode_kernel). */
enum : uint32_t {
	CM_BASE = 0,   /* first row: base phases (accumulator + phase modulation), second: self-modulation amounts */
	CM_INC = 1,    /* first row: phase increments (the chain sums them), second: amounts */
	CM_INLINE = 2, /* no input rows: frequency and amounts are the operator's own lines, evaluated by the feeder wave */
};
enum : uint32_t { CL_FCONST = 1, CL_MUL_GOAL = 2, CL_MUL_HOLD = 4 };
struct ChainDesc {
	uint32_t n;      /* frames to run this segment (0: row pair unused; the other fields are then unset) */
	uint32_t gop;    /* the operator's state (global index) */
	uint32_t wave;
	uint32_t mode;   /* CM_* */
	float coeff;     /* CM_INLINE: 2^32 / srate */
	uint32_t inc_const; /* ... the phase increment when the frequency is one value (CL_FCONST) */
	uint32_t lflags; /* CL_* */
	float mulc;      /* ... multiplier of a ratio line (the parent's frequency, one value) */
	FastLine fl;     /* ... frequency line over the segment */
	FastLine pl;     /* ... self-modulation amount line */
};
static_assert(sizeof(ChainDesc) == 4 * CHAIN_DESC_WORDS && offsetof(ChainDesc, n) == 0, "ChainDesc is 32 dwords, n first");

/* What only a sequential-scan voice needs of a step (FastStep.ramp bit 1): where
 * per-frame frequencies come from and how a ratio line is multiplied. */
enum : uint32_t {
	FA_FVAR_SLOT = 1u << 0, /* ST_OSC: frequency per frame from block buffer freq_off */
	FA_FVAR_LINE = 1u << 1, /* ST_OSC: frequency per frame from its own line `fl` (x multiplier) */
	FA_MUL_GOAL = 1u << 2,  /* the ramp part of the line is a ratio: x multiplier (sau/line.c:72) */
	FA_MUL_HOLD = 1u << 3,  /* the held part of the line is a ratio */
};
struct FastAux {
	uint32_t freq_off, fmul_off; /* block buffers (float offsets) or ~0u */
	float coeff;                 /* 2^32 / srate (wosc.h:30) */
	uint32_t flags;              /* FA_* */
	float mulc;                  /* the multiplier when the parent's frequency is one value */
	uint32_t pad[3];             /* multi-pass voices: [0] index among the voice's running-sum oscillators, [1] its level */
	FastLine fl;                 /* ST_OSC with FA_FVAR_LINE: the frequency line's block */
};
static_assert(sizeof(FastAux) == 80, "FastAux is 20 dwords");


typedef const uint32_t __attribute__((address_space(4))) *const_u32_ptr;
#ifndef FK_GRID
#define FK_GRID 256 /* workgroups at most: one per CU (LDS allows no more at T = 4) */
#endif
#ifndef FK_COMMON
#define FK_COMMON 1
#endif
#ifndef FK_CONSTD
#define FK_CONSTD 1
#endif
#ifndef FK_PREFETCH
#define FK_PREFETCH 0 /* loading the next step early measured 6 % slower (SGPR pressure) */
#endif
__device__ __forceinline__ FastLine load_line_uniform(const FastLine *p) {
	const_u32_ptr q = (const_u32_ptr)(uintptr_t)p;
	union { FastLine s; uint32_t u[12]; } c;
#pragma unroll
	for (int i = 0; i < 12; ++i) c.u[i] = q[i];
	return c.s;
}
__device__ __forceinline__ FastAux load_aux_uniform(const FastAux *p) {
	const_u32_ptr q = (const_u32_ptr)(uintptr_t)p;
	union { FastAux s; uint32_t u[20]; } c;
#pragma unroll
	for (int i = 0; i < 20; ++i) c.u[i] = q[i];
	return c.s;
}
/* value of a ramp at frame t of the segment (lead-in frames t < 0 get the hold value: unused) */
__device__ __forceinline__ float fast_line_value(const FastLine &fl, int t) {
	const uint32_t i = (uint32_t)t;
	return i < fl.goal_len ? sweep_value_inl<true>(fl.sw, i) : fl.hold;
}

/* Between the two passes: the sums of phase increments per row group become
 * exclusive prefixes (what the accumulator has gained before each group). */
__global__ void __launch_bounds__(64) scan_kernel(FastParams P) {
	const uint32_t v = blockIdx.x;
	const int l = threadIdx.x;
	if (P.pass_flags[P.mode - 1] == 0) return;
	const FastInfo fi = P.info[v];
	if (fi.seq != 2 || fi.total == 0) return;
	const uint32_t C = 64u - fi.H;
	const uint32_t nrows = (fi.total + C - 1) / C;
	const uint32_t ngroups = (nrows + P.rows_multi - 1) / P.rows_multi;
	for (uint32_t x = 0; x < fi.n_scan && x < FAST_MAX_SCAN; ++x) {
		if (((fi.lvl_bits >> (2 * x)) & 3u) != P.mode) continue; /* sums of this pass only */
		unsigned long long *a = P.scan + ((size_t)v * FAST_MAX_SCAN + x) * P.scan_groups;
		unsigned long long carry = 0;
		for (uint32_t base = 0; base < ngroups; base += 64) {
			const bool in = base + (uint32_t)l < ngroups;
			const unsigned long long val = in ? a[base + l] : 0ull;
			const unsigned long long incl = wave_incl_scan64_dpp(val);
			if (in) a[base + l] = carry + (incl - val);
			carry += readlane64(incl, 63);
		}
	}
}

/* One step of a voice's plan in immediate form (LDS offsets, constants), so
 * that a row touches no operator records: lazily-constant frequency lines
 * vanish, everything a step needs is 20 dwords in scalar registers. */
__global__ void __launch_bounds__(64) decode_kernel(FastParams P) {
	const uint32_t v = blockIdx.x;
	const int l = threadIdx.x;
	if (P.info[v].total == 0) return;
	/* frames per block buffer: the launch that takes this kind of voice has its own rows per pass */
	const uint32_t NP = 64 * ((P.info[v].seq == 1 || P.info[v].seq == 2) ? P.rows_multi : P.rows);
	const VoiceDesc vd = P.voices[v];
	const uint32_t *ids = P.op_ids + vd.ops_ofs;
	/* lane si handles step si (plan_len <= 64) */
	bool keep = false;
	FastStep f;
	FastLine fl, pl;
	FastAux fa;
	memset(&f, 0, sizeof f); memset(&fl, 0, sizeof fl); memset(&fa, 0, sizeof fa); memset(&pl, 0, sizeof pl);
	bool is_chain = false, chain_line = false, chain_inline = false;
	uint32_t dep = 0;
	const uint32_t seq = P.info[v].seq;
	if ((uint32_t)l < vd.plan_len) {
		const Step *plan = P.steps + vd.plan_ofs;
		/* nesting depth of this step = BEGINs up to and including it minus ENDs before it */
		for (uint32_t q = 0; q <= (uint32_t)l; ++q) {
			const Step sq = plan[q];
			if (sq.flags & SF_BEGIN) ++dep;
			if (q < (uint32_t)l && (sq.flags & SF_END)) --dep;
		}
		const Step st = plan[l];
		const DevOp &o = P.ops[ids[st.op]];
		/* a frequency line is materialised only when it is not one value (sequential-scan voices) */
		keep = !(st.kind == ST_LINE && st.which == L_FREQ && o.rt_fconst_valid);
		/* a W oscillator whose self-modulation is on (generator.c:479-498, wosc.h:273-310): chain_kernel's */
		is_chain = step_is_chain(st, o);
		const uint32_t st_which = st.kind == ST_SMLINE ? (uint32_t)L_PMA : (uint32_t)st.which; /* (ST_SMLINE = the pm_a line into a block) */
		bool zero_fill = false;
		if (o.rt_frozen) {
			/* out of time: of the whole subtree only the root's final step remains,
			 * as a zero fill of its output unless that is layered onto other
			 * modulators' (generator.c:719-728) */
			bool root_end = false;
			if (st.kind == ST_OSC && (st.flags & SF_END)) {
				/* the root is the frozen operator whose enclosing operator (if any) is live */
				uint32_t d2 = 0, frozen_at = 0;
				for (uint32_t q = 0; q <= (uint32_t)l; ++q) {
					const Step sq = plan[q];
					const DevOp &oq = P.ops[ids[sq.op]];
					if (sq.flags & SF_BEGIN) {
						++d2;
						if (!frozen_at && !(oq.flags & OPF_TIME_INF) && oq.time == 0) frozen_at = d2;
					}
					if (q == (uint32_t)l) root_end = (d2 == frozen_at);
					if (sq.flags & SF_END) { if (d2 == frozen_at) frozen_at = 0; --d2; }
				}
			}
			zero_fill = root_end && !(st.flags & SF_LAYER) && !(st.which & OX_VOICE);
			keep = zero_fill;
		}
		/* nesting depth as the row sees it: lanes of extra lead-in the voice has (analyze_kernel)
		 * minus those this operator needs itself -- its values count as defined from lane
		 * H - depth + 1 */
		const uint32_t eff_dep = dep + P.info[v].xlead - min(P.ops[ids[st.op]].st_prev_phase, P.info[v].xlead);
		f.kind = (uint32_t)(st.kind == ST_SMLINE ? (uint8_t)ST_LINE : st.kind) | ((uint32_t)st.flags << 8) | (st_which << 16) | (eff_dep << 24);
		/* block buffers renumbered by liveness (sau_dev_types.h): out, pm, fpm, amp, range end */
		const FastIds cs = P.fast_ids[(seq ? P.ids_full_ofs : 0u) + vd.plan_ofs + l];
		f.out_off = cs.out != NO_SLOT ? (uint32_t)cs.out * NP : ~0u;
		f.pm_off = cs.pm != NO_SLOT ? (uint32_t)cs.pm * NP : ~0u;
		f.fpm_off = cs.fpm != NO_SLOT ? (uint32_t)cs.fpm * NP : ~0u;
		f.amp_off = cs.amp != NO_SLOT ? (uint32_t)cs.amp * NP : ~0u;
		f.aux_off = cs.aux != NO_SLOT ? (uint32_t)cs.aux * NP : ~0u;
		if (st.kind == ST_OSC) f.aux_off = cs.sm != NO_SLOT ? (uint32_t)cs.sm * NP : ~0u; /* self-modulation amounts */
		const uint32_t wv = o.type == OT_WAVE ? (o.wave < 12 ? o.wave : 0) : o.wave;
		f.type = o.type | (wv << 8) | ((o.flags & OPF_OSC_RESET) ? 1u << 16 : 0u);
		f.fc = o.rt_fconst;
		f.inc = rint32w(o.coeff * o.rt_fconst);
		f.phase0 = o.type == OT_NOISE ? o.noise_n : o.phase;
		f.prev_phase = o.type == OT_NOISE ? o.noise_prev : o.prev_phase;
		f.ac = (st.kind == ST_LINE || st.kind == ST_SMLINE) ? o.line[st_which].v0 : o.line[L_AMP].v0;
		f.diff_scale = o.type == OT_WAVE ? P.wc[wv].diff_scale : 0.f;
		f.diff_offset = o.type == OT_WAVE ? P.wc[wv].diff_offset : 0.f;
		f.tab = o.type == OT_WAVE ? P.tab_of_wave[wv] : -1;
		f.gop = ids[st.op];
		f.prev_Is = o.prev_Is;
		if (o.type == OT_RASEG) {
			/* rasg.h:165-222: 64-bit cycle|phase counter, post-increment. The fields a
			 * W oscillator uses for its table and differentiator carry R's options. */
			const bool rate2x = (o.flags & OPF_RATE2X) != 0;
			const unsigned long long inc64 = (unsigned long long)rint64((rate2x ? o.coeff * 2 : o.coeff) * o.rt_fconst);
			f.inc = (uint32_t)inc64;
			f.prev_phase = (uint32_t)(inc64 >> 32);
			f.prev_Is = __longlong_as_double((long long)o.cycle_phase);
			f.tab = (int32_t)((o.ras_func & 0xff) | ((o.ras_flags & 0xffff) << 8) | ((o.wave & 0x7f) << 24));
			f.diff_scale = bits_f(o.ras_level);
			f.diff_offset = bits_f(o.ras_alpha);
			f.type |= rate2x ? 1u << 17 : 0u;
		}
		f.pan = o.line[L_PAN].v0;
		f.ramp = 0;
		if (o.type == OT_AMP) f.fc = 1.f;
		if (is_chain && !zero_fill) {
			uint32_t k = 0; /* its row pair: chains of the voice in plan order, as the host counted them */
			for (uint32_t q = 0; q < (uint32_t)l; ++q) if (step_may_chain(plan[q])) ++k;
			const uint32_t row = vd.chain_base + k;
			f.type |= FT_CHAIN;
			f.pan = bits_f(row);
			ChainDesc cd;
			memset(&cd, 0, sizeof cd);
			cd.n = P.info[v].total; cd.gop = ids[st.op]; cd.wave = wv;
			cd.mode = step_is_chain_acc(st, o) ? CM_INC : CM_BASE;
			if (st.sm == NO_SLOT) { /* the amounts come from the line itself */
				LineState pls = o.line[L_PMA];
				const LineBlock lb = line_begin(pls, P.info[v].total, false, 0.f, lattice_none(), 0);
				pl.sw = lb.sw; pl.goal_len = lb.goal_len; pl.hold = lb.hold; pl.pad = 0;
				chain_line = true;
			}
			uint32_t lstep = ~0u;
			if (step_is_chain_inline(P.chain_inline != 0, plan, (uint32_t)l, ids, P.ops, &lstep)) {
				chain_inline = true;
				cd.mode = CM_INLINE;
				cd.coeff = o.coeff;
				cd.pl = pl;
				cd.mulc = 1.f;
				if (o.rt_fconst_valid) {
					cd.lflags = CL_FCONST;
					cd.inc_const = rint32w(o.coeff * o.rt_fconst);
				} else {
					const Step ls = lstep != ~0u ? plan[lstep] : st;
					const bool have_mul = ls.fmul != NO_SLOT;
					float pf = 1.f;
					if (have_mul && ls.prov != NO_SLOT) pf = P.ops[ids[ls.prov]].rt_fconst;
					LineState fls = o.line[L_FREQ];
					const LineBlock lb = line_begin(fls, P.info[v].total, have_mul, pf, lattice_none(), 0);
					cd.fl.sw = lb.sw; cd.fl.goal_len = lb.goal_len; cd.fl.hold = lb.hold; cd.fl.pad = 0;
					if (lb.mul_goal) cd.lflags |= CL_MUL_GOAL;
					if (lb.mul_hold) cd.lflags |= CL_MUL_HOLD;
					cd.mulc = pf;
				}
			}
			P.chain_desc[row] = cd;
		}
		if (st.kind == ST_OSC && o.type == OT_WAVE && !zero_fill && !is_chain && st.pm == NO_SLOT && st.fpm == NO_SLOT &&
		    o.rt_fconst_valid && f.inc == 0 && ((o.flags & OPF_OSC_RESET) || o.prev_phase == o.phase)) {
			/* Frequency 0, unmodulated: the phase never moves and the differentiator holds its
			 * output (wosc.h:251-252) -- the value it had, or on a restart the one the first
			 * frame computes against phase - one table step (wosc.h:215-231). The step becomes
			 * a constant source; the state the segment leaves behind is known right here. */
			float held = o.prev_s;
			double Is0 = o.prev_Is;
			uint32_t pprev = o.prev_phase;
			if (o.flags & OPF_OSC_RESET) {
				const HerpC23 *g23 = P.g_c23 + (size_t)wv * WAVE_LEN;
				const HerpC01 *g01 = P.g_c01 + (size_t)wv * WAVE_LEN;
				const uint32_t pa = o.phase, pb = o.phase - SLEN;
				Is0 = herp_poly(g23[pa >> SLEN_BITS], g01[pa >> SLEN_BITS], pa);
				const double IsP = herp_poly(g23[pb >> SLEN_BITS], g01[pb >> SLEN_BITS], pb);
				held = wosc_diff(Is0, IsP, (int32_t)SLEN, P.wc[wv].diff_scale, P.wc[wv].diff_offset);
				pprev = pa;
			}
			DevOp &ow = P.ops[ids[st.op]];
			ow.st_phase = o.phase; ow.st_prev_phase = pprev; ow.st_prev_Is = Is0; ow.st_prev_s = held;
			f.type = OT_AMP;
			f.fc = held;
		}
		if (zero_fill) { /* becomes a constant line step */
			f.kind = (uint32_t)ST_LINE | ((uint32_t)L_AMP << 16) | (dep << 24);
			f.ac = 0.f;
			/* every sum pass runs it: whatever reads its buffer there must find the zeros (the
			 * backward data-flow below does not look inside subtrees that are out of time) */
			f.ramp = ((4u << FAST_MAX_LEVELS) - 4u) | FR_CHAIN_IN;
		} else {
			const bool line_step = st.kind == ST_LINE || st.kind == ST_SMLINE;
			const bool amp_inline = st.kind == ST_OSC && st.amp == NO_SLOT;
			/* multiplier of a ratio line: the parent's frequency, one value or a block */
			const bool have_mul = st.fmul != NO_SLOT;
			bool pconst = false; float pf = 1.f;
			if (have_mul && st.prov != NO_SLOT) {
				const DevOp &po = P.ops[ids[st.prov]];
				pconst = po.rt_fconst_valid != 0; pf = po.rt_fconst;
			}
			fa.freq_off = ~0u; fa.fmul_off = ~0u; fa.coeff = o.coeff; fa.flags = 0; fa.mulc = 1.f;
			fa.pad[0] = fa.pad[1] = fa.pad[2] = 0;
			fa.fl.goal_len = 0; fa.fl.hold = 0.f; fa.fl.pad = 0;
			fa.fl.sw = sweep_setup(LN_sah, 0.f, 0.f, 0, 1);
			LineState ls = o.line[line_step ? st_which : L_AMP];
			if (line_step || amp_inline) {
				if (ls.flags & LP_GOAL) {
					const LineBlock lb = line_begin(ls, P.info[v].total, line_step && have_mul, pconst ? pf : 1.f, lattice_none(), 0);
					fl.sw = lb.sw; fl.goal_len = lb.goal_len; fl.hold = lb.hold; fl.pad = 0;
					f.ramp = 1;
					if (lb.mul_goal) fa.flags |= FA_MUL_GOAL;
					if (lb.mul_hold) fa.flags |= FA_MUL_HOLD;
				} else if (line_step && have_mul && (ls.flags & LP_STATE_RATIO)) {
					fa.flags |= FA_MUL_HOLD;
				}
				if (fa.flags & (FA_MUL_GOAL | FA_MUL_HOLD)) {
					if (pconst) fa.mulc = pf; else fa.fmul_off = cs.fmul != NO_SLOT ? (uint32_t)cs.fmul * NP : ~0u;
					f.ramp |= 2;
				}
			}
			const bool is_osc = o.type == OT_WAVE || o.type == OT_RASEG;
			if (seq == 2) {
				/* Pass 1 of a two-pass voice only runs what the phase increments need.
				 * Backward data-flow over the compact block buffers: a step is needed if it
				 * writes a buffer some later needed step (or a running-sum oscillator's
				 * frequency input) reads. */
				unsigned long long want[FAST_MAX_LEVELS];
				bool mine[FAST_MAX_LEVELS];
				for (uint32_t p = 0; p < FAST_MAX_LEVELS; ++p) { want[p] = 0; mine[p] = false; }
				unsigned long long want_c = 0; /* the chain-input pass: what the chains' inputs need */
				bool mine_c = false;
				bool ran_full = false; /* some sum pass evaluates this step in full (and stages its end-of-segment state) */
				unsigned long long want_f = 0; /* the final pass: what the voice's output needs, chains read from their rows */
				bool mine_f = false;
				uint32_t xi = 0;
				for (uint32_t q = vd.plan_len; q-- > 0;) {
					const Step sq = plan[q];
					const DevOp &oq = P.ops[ids[sq.op]];
					if (oq.rt_frozen) continue;
					if (sq.kind == ST_LINE && sq.which == L_FREQ && oq.rt_fconst_valid) continue; /* dropped */
					const FastIds cq = P.fast_ids[P.ids_full_ofs + vd.plan_ofs + q];
					auto bit = [](uint8_t id) -> unsigned long long { return id != NO_SLOT ? 1ull << id : 0ull; };
					const bool q_fvar = sq.kind == ST_OSC && (oq.type == OT_WAVE || oq.type == OT_RASEG) && !oq.rt_fconst_valid &&
						!step_is_chain_acc(sq, oq);
					const uint32_t q_level = q_fvar ? oq.rt_fblk_valid : 0u; /* analyze_kernel left the level there */
					if (q_fvar && q < (uint32_t)l) ++xi;
					const bool writes = sq.kind == ST_LINE || sq.kind == ST_LERP || sq.kind == ST_SMLINE ||
						(sq.kind == ST_OSC && !(sq.which & OX_VOICE));
					const bool rmw = sq.kind == ST_LERP || (sq.kind == ST_OSC && (sq.flags & SF_LAYER));
					{
						const bool q_chain = step_is_chain(sq, oq);
						uint32_t q_ls = ~0u;
						const bool q_inline = q_chain && step_is_chain_inline(P.chain_inline != 0, plan, q, ids, P.ops, &q_ls);
						bool needed = false;
						if (q_inline) {
							/* its inputs are its own lines: the feeder wave of chain_kernel evaluates them */
						} else if (q_chain) {
							needed = true; /* writes its inputs to the rows, nothing else */
							want_c |= bit(cq.freq) | bit(cq.fmul) | bit(cq.pm) | bit(cq.fpm) | bit(cq.sm);
						} else if (writes && (want_c & bit(cq.out))) {
							needed = true;
							if (!rmw) want_c &= ~bit(cq.out);
							want_c |= bit(cq.pm) | bit(cq.fpm) | bit(cq.amp) | bit(cq.aux) | bit(cq.freq) | bit(cq.fmul) | bit(cq.sm);
						}
						if (q == (uint32_t)l) mine_c = needed;
						bool needed_f = sq.kind == ST_VOICE || (sq.kind == ST_OSC && (sq.which & OX_VOICE));
						if (!needed_f && writes && (want_f & bit(cq.out))) {
							needed_f = true;
							if (!rmw) want_f &= ~bit(cq.out);
						}
						/* a running-sum oscillator whose increments are saved (same rule as where the rows are assigned) */
						const bool q_saved = q_fvar && seq == 2 && P.inc_rows && vd.n_inc && sq.fpm == NO_SLOT && !q_chain;
						if (needed_f) {
							if (sq.kind == ST_VOICE) want_f |= bit(cq.out) | bit(cq.pm);
							else if (q_chain) want_f |= bit(cq.amp);
							else if (q_saved) want_f |= bit(cq.pm) | bit(cq.amp) | bit(cq.aux) | bit(cq.sm);
							else want_f |= bit(cq.pm) | bit(cq.fpm) | bit(cq.amp) | bit(cq.aux) | bit(cq.freq) | bit(cq.fmul) | bit(cq.sm);
						}
						if (q == (uint32_t)l) mine_f = needed_f;
					}
					for (uint32_t p = 0; p < FAST_MAX_LEVELS; ++p) { /* sum pass p + 1 */
						bool needed = false;
						if (q_fvar && q_level == p + 1) {
							needed = true; /* as a sums-only step */
							want[p] |= bit(cq.freq) | bit(cq.fmul);
						} else if (!(q_fvar && q_level > p + 1)) {
							/* an ordinary producer (running sums of lower levels have their prefixes by now) */
							if (writes && (want[p] & bit(cq.out))) {
								needed = true;
								if (!rmw) want[p] &= ~bit(cq.out);
								want[p] |= bit(cq.pm) | bit(cq.fpm) | bit(cq.amp) | bit(cq.aux) | bit(cq.freq) | bit(cq.fmul);
							}
						}
						if (q == (uint32_t)l) {
							mine[p] = needed;
							if (needed && !(q_fvar && q_level == p + 1) && p < P.sum_levels) ran_full = true;
						}
					}
				}
				for (uint32_t p = 0; p < FAST_MAX_LEVELS; ++p) if (mine[p]) f.ramp |= 4u << p;
				if (mine_c) f.ramp |= FR_CHAIN_IN;
				/* what only chains' inputs needed has run (and staged its state) in the chain-input pass */
				/* ... and so has what only running sums needed, in their sum passes, when the final pass reads the saved
				 * increments; lines and range blends carry no state of their own */
				if (!mine_f && (mine_c || ran_full || st.kind == ST_LINE || st.kind == ST_SMLINE || st.kind == ST_LERP))
					f.ramp |= FR_FINAL_SKIP;
				fa.pad[1] = (o.type == OT_WAVE || o.type == OT_RASEG) && !o.rt_fconst_valid && !step_is_chain_acc(st, o) ? o.rt_fblk_valid : 0u;
				fa.pad[0] = xi;
			}
			if (seq == 3 && st.kind == ST_OSC && is_osc && !o.rt_fconst_valid) {
				/* single-pass voice: which of its look-back arrays this oscillator has */
				uint32_t xi = 0;
				for (uint32_t q = 0; q < (uint32_t)l; ++q) {
					const Step sq = plan[q];
					const DevOp &oq = P.ops[ids[sq.op]];
					if (oq.rt_frozen) continue;
					if (sq.kind == ST_OSC && (oq.type == OT_WAVE || oq.type == OT_RASEG) && !oq.rt_fconst_valid) ++xi;
				}
				fa.pad[0] = xi; fa.pad[1] = 0;
			}
			if (st.kind == ST_OSC && is_osc && !o.rt_fconst_valid) {
				/* frequency per frame: from its block, or from its own line when it has no block */
				if (st.freq != NO_SLOT) {
					fa.flags |= FA_FVAR_SLOT;
					fa.freq_off = cs.freq != NO_SLOT ? (uint32_t)cs.freq * NP : ~0u;
				} else {
					LineState fls = o.line[L_FREQ];
					const LineBlock lb = line_begin(fls, P.info[v].total, have_mul, pconst ? pf : 1.f, lattice_none(), 0);
					fa.fl.sw = lb.sw; fa.fl.goal_len = lb.goal_len; fa.fl.hold = lb.hold;
					fa.flags |= FA_FVAR_LINE;
					if (lb.mul_goal) fa.flags |= FA_MUL_GOAL;
					if (lb.mul_hold) fa.flags |= FA_MUL_HOLD;
					if (fa.flags & (FA_MUL_GOAL | FA_MUL_HOLD)) {
						if (pconst) fa.mulc = pf; else fa.fmul_off = cs.fmul != NO_SLOT ? (uint32_t)cs.fmul * NP : ~0u;
					}
				}
				f.ramp |= 2;
				if (step_is_chain_acc(st, o)) fa.pad[2] = 1; /* increments to the chain's row, no sums */
				else if (seq == 2 && P.inc_rows && vd.n_inc && st.fpm == NO_SLOT && !is_chain) {
					/* its increments are saved by the sum pass of its level and read back by the final pass
					 * (not with frequency-scaled PM: that needs the frequency itself) */
					uint32_t k = 0;
					for (uint32_t q = 0; q < (uint32_t)l; ++q) {
						const Step sq = plan[q];
						const uint32_t tq = P.ops[ids[sq.op]].type;
						if (sq.kind == ST_OSC && (tq == OT_WAVE || tq == OT_RASEG)) ++k;
					}
					if (k < vd.n_inc) fa.pad[2] = 2u | ((vd.inc_base + k) << 8);
				}
			}
		}
	}
	/* step lists: one per pass that runs the step (a multi-pass voice), else just list 0 */
#pragma unroll
	for (uint32_t li = 0; li < FAST_LISTS; ++li) {
		bool in;
		if (li == 0) in = keep && !(f.ramp & FR_FINAL_SKIP);
		else if (li == 4) in = keep && seq == 2 && (f.ramp & FR_CHAIN_IN);
		else in = keep && seq == 2 && (f.ramp & (2u << li));
		const unsigned long long m = __ballot(in);
		if (in) {
			const uint32_t pos = (uint32_t)__popcll(m & ((1ull << l) - 1ull));
			const size_t at = ((size_t)li * P.n_voices + v) * P.max_steps + pos;
			P.fsteps[at] = f;
			if (f.ramp & 1) P.flines[at] = fl;
			if (f.ramp & 2) P.faux[at] = fa;
			if (chain_line && P.fplines) P.fplines[at] = pl;
		}
		if (l == 0) { if (li == 0) P.info[v].n_fsteps = (uint32_t)__popcll(m); else P.info[v].n_pass[li - 1] = (uint32_t)__popcll(m); }
	}
	/* chains the chain-input pass has to feed (the others are fed by chain_kernel's own feeder wave) */
	const unsigned long long mc = __ballot(is_chain && keep && !chain_inline);
	if (l == 0) P.info[v].n_chain = (uint32_t)__popcll(mc);
}

/* lane l receives lane l-1's value (lane 0: zero; it is lead-in) */
__device__ __forceinline__ uint32_t lane_prev(uint32_t x) {
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ double lane_prev(double x) {
	const uint32_t lo = lane_prev((uint32_t)__double2loint(x));
	const uint32_t hi = lane_prev((uint32_t)__double2hiint(x));
	return __hiloint2double((int)hi, (int)lo);
}

/* Uniform (scalar-cache) load of one decoded step: the address is the same
 * for the whole wave and the memory was written by an earlier kernel. */
__device__ __forceinline__ FastStep load_step_uniform(const FastStep *p) {
	const_u32_ptr q = (const_u32_ptr)(uintptr_t)p;
	union { FastStep s; uint32_t u[20]; } c;
#pragma unroll
	for (int i = 0; i < 20; ++i) c.u[i] = q[i];
	return c.s;
}

/* (A 24-byte interleaved LDS entry read with ds_read2_b64 + ds_read_b64 was
 * measured 60 % worse in bank conflicts than this split 16 + 8 byte layout.) */
/* One voice's share of this wave's work. SCAN: built with the running-sum code
 * (frequency ramps, FM); the plain build stays as lean as closed-form voices
 * need it (the same code with the running-sum branches compiled in was 27 %
 * slower on them), and a kernel that may meet both kinds holds both copies. */
template <int T, int SCAN, bool REPAIR = false>
__device__ __forceinline__ void fast_voice(const FastParams &P, const uint32_t v, const FastInfo &fi,
		float *slots, unsigned long long *carry, const HerpC23 *t23, const HerpC01 *t01, const int l,
		const uint32_t wpv, const uint32_t cstart, unsigned long long *lring = nullptr) {
	constexpr int NP = 64 * T;
	(void)NP;
	const uint32_t fast_total = uni(fi.total);
	if (fast_total == 0) return;
	const VoiceDesc vd = P.voices[v];
	const uint32_t H = uni(fi.H);

	/* this pass's own list of the voice's decoded steps */
	/* SCAN: 0 closed-form phases only; 1 every kind of running-sum voice; 2 single-pass (look-back) voices only,
	 * without the code of the several-pass forms and the feedback chains */
	constexpr bool FULL = SCAN == 1;
	const uint32_t li = FULL ? fast_list_of(P.mode, P.sum_levels) : 0u;
	const uint32_t n_fsteps = uni(li == 0 ? fi.n_fsteps : li == 1 ? fi.n_pass[0] : li == 2 ? fi.n_pass[1] : li == 3 ? fi.n_pass[2] : fi.n_pass[3]);
	const size_t list_at = ((size_t)li * P.n_voices + v) * P.max_steps;
	const FastStep *fsteps = P.fsteps + list_at;
	const FastLine *flines = P.flines + list_at;
	const FastAux *faux = P.faux + list_at;
	const FastLine *fplines = P.fplines ? P.fplines + list_at : nullptr;
	/* sequential-scan voices: the one wave with cstart == 0 walks every row group in order */
	const uint32_t seq_kind = SCAN ? uni(fi.seq) : 0u;
	const bool seq = FULL && seq_kind == 1;    /* one wave, in order */
	const bool two = FULL && seq_kind == 2;    /* two passes, every wave */
	const bool look = SCAN == 2;               /* one pass, every wave, prefixes by look-back (a build of its own) */
	unsigned long long *lookv = look ? P.look + (size_t)vd.look_base * 2 * P.scan_groups : nullptr;
	/* the single-pass build: a voice with one wave carries its sums in LDS like an in-order voice; the waves of
	 * one workgroup look back through rings in LDS; voices spread wider go through HBM */
	const bool look_own = SCAN == 2 && wpv == 1;
	const bool look_lds = SCAN == 2 && lring && wpv >= 2 && wpv <= 16 && (16 % wpv) == 0;
	const uint32_t lk_ring = 4 * wpv;
	unsigned long long *lk_base = look_lds ? lring + (uni((uint32_t)threadIdx.x >> 6) / wpv) * lk_ring : nullptr;
	if (seq && cstart != 0) return;
	const uint32_t gstride = seq ? 1u : wpv;
	unsigned long long *scan = two ? P.scan + (size_t)v * FAST_MAX_SCAN * P.scan_groups : nullptr;
	float *vrow = P.vout + (size_t)vd.out_row * P.row_stride;
	float *prow = (vd.pan_dynamic_row != ~0u) ? P.pan + (size_t)vd.pan_dynamic_row * P.row_stride : nullptr;
	/* A wave renders T rows at a time; a row is 64 consecutive frames, one per
	 * lane, of which the first H are lead-in (recomputed, not stored). */
	const uint32_t C = 64u - H;                       /* new frames per row */
	const uint32_t nrows = (fast_total + C - 1) / C;
	const uint32_t ngroups = (nrows + T - 1) / T;
	const uint32_t last_group = ((fast_total - 1) / C) / T; /* holds the segment's last frame */
	uint32_t zero_acc = 0; /* nonzero: some hold-previous run could not be resolved here */

	uint32_t *const rep = P.repair + (size_t)v * FAST_REPAIR_WORDS;
	/* REPAIR: the noted row groups instead of all, each evaluated FAST_REPAIR_SHIFT frames early */
	uint32_t n_iter = REPAIR ? min(uni(rep[0]), FAST_MAX_REPAIR) : ngroups;
	uint32_t it_lo = 0;
	if (!REPAIR && SCAN != 2 && P.range_mode != 0) { /* (the single-pass build never runs in chunks) */
		if (seq) { if (!P.range_last) return; } /* one wave in order, carries in LDS: in the last chunk's launch, all of it */
		else {
			const uint32_t tc = (uint32_t)T * C;
			if (P.range_mode == 1) { /* groups that start in [f_lo, f_hi) */
				it_lo = (P.f_lo + tc - 1) / tc;
				n_iter = min(ngroups, P.f_hi > 0xffffffffu - tc ? ngroups : (P.f_hi + tc - 1) / tc);
			} else { /* groups that end in (f_lo, f_hi]; the last one ends with the segment */
				it_lo = P.f_lo / tc;
				n_iter = fast_total <= P.f_hi ? ngroups : min(ngroups, P.f_hi / tc);
				if (P.f_lo >= fast_total) n_iter = 0;
			}
		}
	}
	for (uint32_t it = it_lo + cstart; it < n_iter; it += gstride) {
		const uint32_t cg = REPAIR ? uni(rep[2 + 2 * it]) : it;
		const uint32_t repair_rows = REPAIR ? uni(rep[3 + 2 * it]) : 0u;
		const int t0 = (int)(cg * T * C) - (int)H + l - (REPAIR ? (int)FAST_REPAIR_SHIFT : 0); /* this lane's frame in row 0 */
		const bool first_group = (cg == 0);
		const bool is_last_group = (cg == last_group);
		uint32_t held_rows = 0; /* rows with a hold this evaluation could not resolve */
		(void)repair_rows;
#if FK_PREFETCH
		FastStep fnext = load_step_uniform(fsteps);
#endif
		for (uint32_t si = 0; si < n_fsteps; ++si) {
#if FK_PREFETCH
			const FastStep f = fnext;
			fnext = load_step_uniform(fsteps + (si + 1 < n_fsteps ? si + 1 : si)); /* in flight during this step */
#else
			const FastStep f = load_step_uniform(fsteps + si);
#endif
			const uint32_t kind = f.kind & 0xff;
			const uint32_t flags = (f.kind >> 8) & 0xff;
			const bool sum_pass = FULL && P.mode != 0 && P.mode <= P.sum_levels;
			if (sum_pass && !(f.ramp & (2u << P.mode))) continue; /* not needed for this pass's phase increments */
			const bool chain_in = FULL && P.mode == P.sum_levels + 2; /* the pass that writes the chains' inputs */
			if (chain_in && !(f.ramp & FR_CHAIN_IN)) continue;
			if (FULL && P.mode == P.sum_levels + 1 && (f.ramp & FR_FINAL_SKIP)) continue;
			if (kind == ST_OSC) {
				const uint32_t type = f.type & 0xff;
				const bool wave_env = (flags & SF_WAVE_ENV) != 0;
				const bool layer = (flags & SF_LAYER) != 0;
				const bool to_voice = ((f.kind >> 16) & OX_VOICE) != 0;
				float s[T];
				const bool chain = FULL && (f.type & FT_CHAIN) != 0;
				if (type == OT_WAVE && chain && P.mode == P.sum_levels + 1) { /* (the final pass) */
					/* a feedback chain: chain_kernel has run it; its samples are in the row */
					const float *crow = P.chain_rows + (size_t)2 * f_bits(f.pan) * P.chain_stride;
#pragma unroll
					for (int k = 0; k < T; ++k) {
						const int t = t0 + k * (int)C;
						s[k] = (t >= 0 && t < (int)fast_total) ? crow[t] : 0.f;
					}
				} else if (type == OT_WAVE) {
					const bool has_pm = f.pm_off != ~0u, has_fpm = f.fpm_off != ~0u;
					/* this operator's values are defined from lane p_min on
					 * (one more lead-in sample per nesting level below it) */
					const int p_min = (int)H - (int)(f.kind >> 24) + 1;
					bool done = false;
					if (FK_COMMON && f.tab >= 0 && !has_fpm && !first_group && !is_last_group && !(f.ramp & 2) && !chain) {
						/* the common case, straight-line: table in LDS, plain PM or
						 * none, no segment edge in this group */
						uint32_t ph[T];
						{
							uint32_t acc = f.phase0 + f.inc * (uint32_t)(t0 + 1);
							const uint32_t row_inc = f.inc * C;
#pragma unroll
							for (int k = 0; k < T; ++k) { ph[k] = acc; acc += row_inc; }
						}
						bool ok = true;
						if (has_pm) {
							float pm[T];
							bool big = false;
#pragma unroll
							for (int k = 0; k < T; ++k) {
								pm[k] = slots[f.pm_off + k * 64];
								big |= !(fabsf(pm[k]) < 0x1p20f);
							}
							ok = !__any(big);
#pragma unroll
							for (int k = 0; k < T; ++k) ph[k] += rint32w_p31_small(pm[k]);
						}
						if (ok) {
							const HerpC23 *l23 = t23 + (size_t)f.tab * WAVE_LEN;
							const HerpC01 *l01 = t01 + (size_t)f.tab * WAVE_LEN;
							double Is[T];
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const uint32_t ind = ph[k] >> SLEN_BITS;
								Is[k] = herp_poly(l23[ind], l01[ind], ph[k]);
							}
							if (FK_CONSTD && !has_pm && f.inc != 0) {
								/* unmodulated: every phase step is inc, one division serves all */
								const double x = (double)div_f32_normal(f.diff_scale, (float)(int32_t)f.inc);
#pragma unroll
								for (int k = 0; k < T; ++k)
									s[k] = (float)((Is[k] - lane_prev(Is[k])) * x + (double)f.diff_offset);
								done = true;
							} else {
								bool zero = false;
#pragma unroll
								for (int k = 0; k < T; ++k) {
									const int32_t d = (int32_t)(ph[k] - lane_prev(ph[k]));
									zero |= (d == 0);
									s[k] = wosc_diff(Is[k], lane_prev(Is[k]), d, f.diff_scale, f.diff_offset);
								}
								done = !__any(zero && l >= p_min);
							}
						}
					}
					if (!done) {
						uint32_t ph[T];
						double Is[T];
						float fv[T]; /* frequency per frame (freq-scaled PM reads it) */
						bool fvar = false;
						if (SCAN && (f.ramp & 2)) {
							/* the frequency varies (ramp, FM): phase is a running sum of per-frame
							 * increments (wosc.h:135-169). This wave walks the voice's rows in order;
							 * `carry` holds the accumulator at the frame before each row's new frames. */
							const FastAux fa = load_aux_uniform(faux + si);
							fvar = (fa.flags & (FA_FVAR_SLOT | FA_FVAR_LINE)) != 0;
							if (fvar) {
								uint32_t S[T];
								auto freq_at = [&](int k, int t) -> float { /* the frequency at row k's frame t */
									if (fa.flags & FA_FVAR_SLOT) return slots[fa.freq_off + k * 64];
									float v = fast_line_value(fa.fl, t);
									const bool in_goal = (uint32_t)t < fa.fl.goal_len;
									if (fa.flags & (in_goal ? FA_MUL_GOAL : FA_MUL_HOLD))
										v *= fa.fmul_off != ~0u ? slots[fa.fmul_off + k * 64] : fa.mulc;
									return v;
								};
								/* saved increments (FastParams.inc_rows): written by the sum pass of this oscillator's
								 * level, read back by the final pass in place of the frequency */
								uint32_t *irow = (FULL && (fa.pad[2] & 2u)) ? P.inc_rows + (size_t)2 * (fa.pad[2] >> 8) * P.inc_stride : nullptr;
								const bool inc_read = irow && P.mode == P.sum_levels + 1;
								const bool inc_write = irow && two && P.mode == fa.pad[1];
								if (FULL) {
#pragma unroll
									for (int k = 0; k < T; ++k) {
										const int t = t0 + k * (int)C;
										uint32_t r;
										if (inc_read) {
											r = (t >= 0 && t < (int)fast_total) ? irow[t] : 0u;
											fv[k] = 0.f; /* (only frequency-scaled PM reads it, and such oscillators save nothing) */
										} else {
											const float v = freq_at(k, t);
											fv[k] = v;
											const float x = fa.coeff * v;
											/* llrintf(x) mod 2^32 (wosc.h:145): adding 1.5 * 2^52 in f64 rounds to the nearest
											 * integer and leaves it in the low word; exact while |x| < 2^51 */
											r = fabsf(x) < 0x1p50f ? (uint32_t)__double2loint((double)x + 0x1.8p52) : rint32w(x);
											if (inc_write && l >= (int)H && t >= 0 && t < (int)fast_total) irow[t] = r;
										}
										const uint32_t inc = (t >= 0 && t < (int)fast_total) ? r : 0u;
										if (chain && fa.pad[2]) S[k] = inc; /* chain_kernel does the summing */
										else S[k] = wave_incl_scan_dpp(inc);
									}
								} else { /* the single-pass build: one test per group for the rounding form */
									float x[T];
									bool big = false;
#pragma unroll
									for (int k = 0; k < T; ++k) {
										fv[k] = freq_at(k, t0 + k * (int)C);
										x[k] = fa.coeff * fv[k];
										big |= !(fabsf(x[k]) < 0x1p50f);
									}
									uint32_t r[T];
									if (!__any(big)) {
#pragma unroll
										for (int k = 0; k < T; ++k) r[k] = (uint32_t)__double2loint((double)x[k] + 0x1.8p52);
									} else {
#pragma unroll
										for (int k = 0; k < T; ++k) r[k] = rint32w(x[k]);
									}
#pragma unroll
									for (int k = 0; k < T; ++k) {
										const int t = t0 + k * (int)C;
										S[k] = wave_incl_scan_dpp((t >= 0 && t < (int)fast_total) ? r[k] : 0u);
									}
								}
								if (chain && fa.pad[2]) {
									/* chain-input pass of a chain that accumulates its own phase: increments and amounts */
									float *brow = P.chain_rows + (size_t)2 * f_bits(f.pan) * P.chain_stride;
									float *arow = brow + P.chain_stride;
									FastLine pl;
									const bool from_line = f.aux_off == ~0u;
									if (from_line) pl = load_line_uniform(fplines + si);
#pragma unroll
									for (int k = 0; k < T; ++k) {
										const int t = t0 + k * (int)C;
										const float a = from_line ? fast_line_value(pl, t) : slots[f.aux_off + k * 64];
										if (l >= (int)H && t >= 0 && t < (int)fast_total) {
											((u32_alias *)brow)[t] = S[k];
											arow[t] = a;
										}
									}
									continue;
								}
								/* the accumulator at the frame before this group's first new frame: carried by
								 * this wave (in-order voices), or the prefix of all earlier groups' sums */
								unsigned long long *sums = two ? scan + (size_t)fa.pad[0] * P.scan_groups : nullptr;
								const bool sum_me = two && P.mode == fa.pad[1]; /* this pass computes this oscillator's sums */
								uint32_t acc;
								if (look_own) {
									acc = first_group ? f.phase0 : (uint32_t)carry[si];
								} else if (look) {
									uint32_t tot = 0;
#pragma unroll
									for (int k = 0; k < T; ++k)
										tot += (uint32_t)__builtin_amdgcn_readlane((int)S[k], 63) - (uint32_t)__builtin_amdgcn_readlane((int)S[k], (int)H - 1);
									acc = f.phase0 + (look_lds ? lookback32<true>(lk_base + (size_t)fa.pad[0] * 2 * 64, cg, tot, 0, lk_ring, l)
									                           : lookback32<false>(lookv + (size_t)fa.pad[0] * 2 * P.scan_groups, cg, tot, P.look_epoch, 0, l));
								} else {
									acc = two ? (sum_me ? 0u : f.phase0 + (uint32_t)sums[cg])
									          : (first_group ? f.phase0 : (uint32_t)carry[si]);
								}
#pragma unroll
								for (int k = 0; k < T; ++k) {
									const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)S[k], (int)H - 1);
									const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)S[k], 63);
									ph[k] = acc + (S[k] - lead);
									acc += last - lead;
								}
								if (two) {
									if (sum_me) { /* this pass ends here for this oscillator */
										if (l == 0) sums[cg] = (unsigned long long)acc;
										continue;
									}
								} else if ((!look || look_own) && l == 0) {
									carry[si] = (unsigned long long)acc;
								}
							}
						}
						if (!fvar) {
							/* phase0 + inc*(t+1): one multiply per lane, then adds */
							uint32_t acc = f.phase0 + f.inc * (uint32_t)(t0 + 1);
							const uint32_t row_inc = f.inc * C;
#pragma unroll
							for (int k = 0; k < T; ++k) { ph[k] = acc; acc += row_inc; fv[k] = f.fc; }
						}
						if (SCAN && is_last_group) { /* the accumulator after the segment's last frame, before modulation */
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)C;
								if (t == (int)fast_total - 1 && l >= (int)H) P.ops[f.gop].st_phase = ph[k];
							}
						}
						if (has_pm && !has_fpm) {
							float pm[T];
							bool big = false;
#pragma unroll
							for (int k = 0; k < T; ++k) {
								pm[k] = slots[f.pm_off + k * 64];
								big |= !(fabsf(pm[k]) < 0x1p20f);
							}
							if (!__any(big)) {
#pragma unroll
								for (int k = 0; k < T; ++k) ph[k] += rint32w_p31_small(pm[k]);
							} else {
#pragma unroll
								for (int k = 0; k < T; ++k) ph[k] += rint32w_p31(pm[k]);
							}
						} else if (has_pm || has_fpm) {
							float pm[T], fpm[T];
#pragma unroll
							for (int k = 0; k < T; ++k) { pm[k] = 0.f; fpm[k] = 0.f; }
							if (has_pm) {
#pragma unroll
								for (int k = 0; k < T; ++k) pm[k] = slots[f.pm_off + k * 64];
							}
							if (has_fpm) {
#pragma unroll
								for (int k = 0; k < T; ++k) fpm[k] = slots[f.fpm_off + k * 64];
							}
							if (has_pm) {
#pragma unroll
								for (int k = 0; k < T; ++k) ph[k] += pm_offset32(true, true, pm[k], fpm[k], fv[k]);
							} else {
#pragma unroll
								for (int k = 0; k < T; ++k) ph[k] += pm_offset32(false, true, 0.f, fpm[k], fv[k]);
							}
						}
						if (chain) {
							/* chain-input pass: base phases (accumulator + phase modulation; the feedback term is
							 * chain_kernel's) and self-modulation amounts to the chain's rows, nothing else */
							float *brow = P.chain_rows + (size_t)2 * f_bits(f.pan) * P.chain_stride;
							float *arow = brow + P.chain_stride;
							FastLine pl;
							const bool from_line = f.aux_off == ~0u;
							if (from_line) pl = load_line_uniform(fplines + si);
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)C;
								const float a = from_line ? fast_line_value(pl, t) : slots[f.aux_off + k * 64];
								if (l >= (int)H && t >= 0 && t < (int)fast_total) {
									((u32_alias *)brow)[t] = ph[k];
									arow[t] = a;
								}
							}
							continue;
						}
						const bool reset = (f.type >> 16) & 1;
						if (first_group) {
							/* t = -1: the sample before the segment (wosc.h:215-231 on restart) */
							const uint32_t nxt = __shfl_down(ph[0], 1);
							if (l == (int)H - 1) ph[0] = reset ? nxt - SLEN : f.prev_phase;
						}
						if (f.tab >= 0) {
							const HerpC23 *l23 = t23 + (size_t)f.tab * WAVE_LEN;
							const HerpC01 *l01 = t01 + (size_t)f.tab * WAVE_LEN;
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const uint32_t ind = ph[k] >> SLEN_BITS;
								Is[k] = herp_poly(l23[ind], l01[ind], ph[k]);
							}
						} else {
							const uint32_t wave = (f.type >> 8) & 0xff;
							const HerpC23 *g23 = P.g_c23 + (size_t)wave * WAVE_LEN;
							const HerpC01 *g01 = P.g_c01 + (size_t)wave * WAVE_LEN;
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const uint32_t ind = ph[k] >> SLEN_BITS;
								Is[k] = herp_poly(g23[ind], g01[ind], ph[k]);
							}
						}
						if (first_group && !reset) {
							if (l == (int)H - 1) Is[0] = f.prev_Is;
						}
						uint32_t pph[T];
						bool zero = false;
						if (FK_CONSTD && !has_pm && !has_fpm && !first_group && f.inc != 0 && !fvar) {
							/* unmodulated: every phase step is inc, one division serves all */
							const double x = (double)div_f32_normal(f.diff_scale, (float)(int32_t)f.inc);
#pragma unroll
							for (int k = 0; k < T; ++k) {
								pph[k] = ph[k] - f.inc;
								const double pIs = lane_prev(Is[k]);
								s[k] = (float)((Is[k] - pIs) * x + (double)f.diff_offset);
							}
						} else {
#pragma unroll
							for (int k = 0; k < T; ++k) {
								pph[k] = lane_prev(ph[k]);
								const double pIs = lane_prev(Is[k]);
								const int32_t d = (int32_t)(ph[k] - pph[k]);
								zero |= (d == 0);
								s[k] = wosc_diff(Is[k], pIs, d, f.diff_scale, f.diff_offset);
							}
						}
						if (__any(zero && l >= p_min)) {
							/* dphase == 0: the differentiator holds its previous output
							 * (wosc.h:251-252). Isolated cases resolve inside the row; a
							 * run that reaches back past the lead-in goes to the block loop. */
							bool held[T], src[T]; /* src: holds a defined output to copy from */
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)C;
								const bool defined = l >= p_min && t >= 0;
								held[k] = (ph[k] == pph[k]) && defined && t < (int)fast_total;
								src[k] = defined && !held[k];
							}
							for (int it = 0; it < 64; ++it) {
								bool changed = false;
#pragma unroll
								for (int k = 0; k < T; ++k) {
									const float sp = __shfl_up(s[k], 1);
									const bool okp = __shfl_up(src[k], 1);
									if (held[k] && okp && l > 0) { s[k] = sp; held[k] = false; src[k] = true; changed = true; }
								}
								if (!__any(changed)) break;
							}
#pragma unroll
							for (int k = 0; k < T; ++k) {
								/* running-sum voices have a lane of slack (analyze_kernel): a hold left on the
								 * operator's first defined lane is harmless there */
								if (SCAN && l == p_min) held[k] = false;
								if (held[k]) rep[1] = ((uint32_t)(l - p_min) << 24) | ((uint32_t)k << 20) | (si << 12) | (cg & 0xfff); /* debug */
								held_rows |= __any(held[k]) ? (1u << k) : 0u;
							}
						}
						if (is_last_group) {
							/* the row that holds the segment's last frame stages the state */
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)C;
								if (t == (int)fast_total - 1 && l >= (int)H) {
									DevOp &o = P.ops[f.gop];
									o.st_prev_phase = ph[k];
									o.st_prev_Is = Is[k];
									o.st_prev_s = s[k];
								}
							}
						}
					}
				} else if (type == OT_RASEG) {
					/* rasg.h:165-222 + 692-743: frame t reads the counter cp0 + inc * t (+ PM) */
					const bool rate2x = (f.type >> 17) & 1;
					const float phase_scale = rate2x ? 0x1p31f * 2 : 0x1p31f;
					const RasParams rp = ras_params((uint32_t)f.tab & 0xff, ((uint32_t)f.tab >> 8) & 0xffff,
							f_bits(f.diff_scale), f_bits(f.diff_offset), ((uint32_t)f.tab >> 24) & 0x7f);
					const unsigned long long inc64 = ((unsigned long long)f.prev_phase << 32) | f.inc;
					const unsigned long long cp0 = (unsigned long long)__double_as_longlong(f.prev_Is);
					const bool has_pm = f.pm_off != ~0u, has_fpm = f.fpm_off != ~0u;
					unsigned long long cpv[T]; /* the counter each frame reads (post-increment), before PM */
					float fv[T];
					bool fvar = false;
					if (SCAN && (f.ramp & 2)) {
						/* the frequency varies: the counter is a running sum of 64-bit increments */
						const FastAux fa = load_aux_uniform(faux + si);
						fvar = (fa.flags & (FA_FVAR_SLOT | FA_FVAR_LINE)) != 0;
						if (fvar) {
							const float rcoeff = rate2x ? fa.coeff * 2 : fa.coeff;
							unsigned long long S[T], incv[T];
							uint32_t *irow = (FULL && (fa.pad[2] & 2u)) ? P.inc_rows + (size_t)2 * (fa.pad[2] >> 8) * P.inc_stride : nullptr;
							const bool inc_read = irow && P.mode == P.sum_levels + 1;
							const bool inc_write = irow && two && P.mode == fa.pad[1];
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const int t = t0 + k * (int)C;
								const bool in_seg = t >= 0 && t < (int)fast_total;
								if (inc_read) { /* saved by the sum pass of its level: low and high words */
									incv[k] = in_seg ? ((unsigned long long)irow[P.inc_stride + t] << 32) | irow[t] : 0ull;
									fv[k] = 0.f;
								} else {
								float v;
								if (fa.flags & FA_FVAR_SLOT) {
									v = slots[fa.freq_off + k * 64];
								} else {
									v = fast_line_value(fa.fl, t);
									const bool in_goal = (uint32_t)t < fa.fl.goal_len;
									if (fa.flags & (in_goal ? FA_MUL_GOAL : FA_MUL_HOLD))
										v *= fa.fmul_off != ~0u ? slots[fa.fmul_off + k * 64] : fa.mulc;
								}
								fv[k] = v;
								incv[k] = in_seg ? (unsigned long long)rint64(rcoeff * v) : 0ull;
								if (inc_write && l >= (int)H && in_seg) { irow[t] = (uint32_t)incv[k]; irow[P.inc_stride + t] = (uint32_t)(incv[k] >> 32); }
								}
								S[k] = wave_incl_scan64_dpp(incv[k]);
							}
							unsigned long long *sums = two ? scan + (size_t)fa.pad[0] * P.scan_groups : nullptr;
							const bool sum_me = two && P.mode == fa.pad[1];
							unsigned long long acc;
							if (look_own) {
								acc = first_group ? cp0 : carry[si];
							} else if (look) {
								unsigned long long tot = 0;
#pragma unroll
								for (int k = 0; k < T; ++k) tot += readlane64(S[k], 63) - readlane64(S[k], (int)H - 1);
								if (look_lds) {
									unsigned long long *e_lo = lk_base + (size_t)fa.pad[0] * 2 * 64;
									acc = cp0 + lookback64<true>(e_lo, e_lo + 64, cg, tot, 0, lk_ring, l);
								} else {
									unsigned long long *e_lo = lookv + (size_t)fa.pad[0] * 2 * P.scan_groups;
									acc = cp0 + lookback64<false>(e_lo, e_lo + P.scan_groups, cg, tot, P.look_epoch, 0, l);
								}
							} else {
								acc = two ? (sum_me ? 0ull : cp0 + sums[cg])
								          : (first_group ? cp0 : carry[si]);
							}
#pragma unroll
							for (int k = 0; k < T; ++k) {
								const unsigned long long lead = readlane64(S[k], (int)H - 1);
								const unsigned long long last = readlane64(S[k], 63);
								cpv[k] = acc + (S[k] - lead) - incv[k];
								acc += last - lead;
							}
							if (two) {
								if (sum_me) {
									if (l == 0) sums[cg] = acc;
									continue;
								}
							} else if ((!look || look_own) && l == 0) {
								carry[si] = acc;
							}
							if (is_last_group) { /* the counter after the segment's last frame */
#pragma unroll
								for (int k = 0; k < T; ++k) {
									const int t = t0 + k * (int)C;
									if (t == (int)fast_total - 1 && l >= (int)H)
										P.ops[f.gop].st_prev_Is = __longlong_as_double((long long)(cpv[k] + incv[k]));
								}
							}
						}
					}
					if (!fvar) {
#pragma unroll
						for (int k = 0; k < T; ++k) {
							const int t = t0 + k * (int)C;
							cpv[k] = cp0 + inc64 * (unsigned long long)(long long)t;
							fv[k] = f.fc;
						}
					}
#pragma unroll
					for (int k = 0; k < T; ++k) {
						unsigned long long cp = cpv[k];
						if (has_pm || has_fpm)
							cp += (unsigned long long)pm_offset(has_pm, has_fpm,
									has_pm ? slots[f.pm_off + k * 64] : 0.f,
									has_fpm ? slots[f.fpm_off + k * 64] : 0.f, fv[k], phase_scale);
						uint32_t cyc;
						float phf;
						ras_split(cp, cyc, phf);
						s[k] = ras_sample(rp, cyc, phf, true);
					}
				} else if (type == OT_NOISE) {
					const uint32_t nz = (f.type >> 8) & 0xff;
					const uint32_t n0 = f.phase0, nprev = f.prev_phase;
#pragma unroll
					for (int k = 0; k < T; ++k) {
						const int t = t0 + k * (int)C;
						const uint32_t n = n0 + (uint32_t)t;
						if (nz == NZ_vi) {
							uint32_t s1 = ranfast32(n);
							uint32_t s0 = t == 0 ? nprev : ranfast32(n - 1);
							s[k] = fscalei((s1 / 2) - (s0 / 2), 0x1p-31f);
						} else if (nz == NZ_bv) {
							int32_t s1 = noise_bv_term(n);
							int32_t s0 = t == 0 ? (int32_t)nprev : noise_bv_term(n - 1);
							s[k] = (float)(s1 - s0);
						} else {
							s[k] = noise_stateless(nz, n);
						}
					}
				} else { /* OT_AMP (generator.c:517-518: 1), or an oscillator whose output stands still */
#pragma unroll
					for (int k = 0; k < T; ++k) s[k] = f.fc;
				}
				/* amplitude and combine: generator.c:384-440 */
				float r[T];
				if (f.amp_off != ~0u) {
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = slots[f.amp_off + k * 64];
				} else if (f.ramp & 1) { /* amplitude ramp in progress, sau/line.c:65-281 */
					const FastLine fl = load_line_uniform(flines + si);
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = fast_line_value(fl, t0 + k * (int)C);
				} else {
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = f.ac;
				}
				if (layer) {
#pragma unroll
					for (int k = 0; k < T; ++k)
						r[k] = mix_combine(slots[f.out_off + k * 64], s[k], r[k], wave_env, true);
				} else if (wave_env) {
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = mix_combine(0.f, s[k], r[k], true, false);
				} else {
#pragma unroll
					for (int k = 0; k < T; ++k) r[k] = s[k] * r[k];
				}
				if (to_voice) {
#pragma unroll
					for (int k = 0; k < T; ++k) {
						const int t = t0 + k * (int)C;
						const bool mine = REPAIR ? (l == (int)(H + FAST_REPAIR_SHIFT) && ((repair_rows >> k) & 1u)) : (l >= (int)H);
						if (mine && t < (int)fast_total) vrow[t] = r[k];
					}
				} else {
#pragma unroll
					for (int k = 0; k < T; ++k) slots[f.out_off + k * 64] = r[k];
				}
			} else if (kind == ST_LINE) {
				/* held line: v0 (sau/line.c:435-442); ratio lines only exist for freq */
				if (f.ramp) {
					FastLine fl;
					fl.goal_len = 0; fl.hold = f.ac; fl.pad = 0;
					fl.sw = sweep_setup(LN_sah, 0.f, 0.f, 0, 1);
					if (f.ramp & 1) fl = load_line_uniform(flines + si);
					uint32_t mflags = 0, fmul_off = ~0u;
					float mulc = 1.f;
					if (SCAN && (f.ramp & 2)) { /* ratio line: x the parent's frequency (sau/line.c:72) */
						const FastAux fa = load_aux_uniform(faux + si);
						mflags = fa.flags; fmul_off = fa.fmul_off; mulc = fa.mulc;
					}
#pragma unroll
					for (int k = 0; k < T; ++k) {
						const int t = t0 + k * (int)C;
						float v = fast_line_value(fl, t);
						const bool in_goal = (uint32_t)t < fl.goal_len;
						if (mflags & (in_goal ? FA_MUL_GOAL : FA_MUL_HOLD))
							v *= fmul_off != ~0u ? slots[fmul_off + k * 64] : mulc;
						slots[f.out_off + k * 64] = v;
					}
				} else {
#pragma unroll
					for (int k = 0; k < T; ++k) slots[f.out_off + k * 64] = f.ac;
				}
			} else if (kind == ST_LERP) { /* generator.c:466-467 */
#pragma unroll
				for (int k = 0; k < T; ++k) {
					float pv = slots[f.out_off + k * 64];
					pv += (slots[f.aux_off + k * 64] - pv) * slots[f.pm_off + k * 64];
					slots[f.out_off + k * 64] = pv;
				}
			} else if (kind == ST_VOICE) { /* generator.c:749-788 with pan modulators */
#pragma unroll
				for (int k = 0; k < T; ++k) {
					const int t = t0 + k * (int)C;
					const bool mine = REPAIR ? (l == (int)(H + FAST_REPAIR_SHIFT) && ((repair_rows >> k) & 1u)) : (l >= (int)H);
					if (mine && t < (int)fast_total) {
						vrow[t] = slots[f.out_off + k * 64];
						if (prow) prow[t] = f.pm_off != ~0u ? slots[f.pm_off + k * 64] : f.pan;
					}
				}
			}
		}
		if (held_rows) {
			/* to the repair pass -- unless this is it, the group touches an end of the segment
			 * (carried state sits at fixed lanes there) or the voice has running sums */
			bool noted = false;
			if (!REPAIR && !SCAN && P.repair_on && !first_group && !is_last_group &&
			    (int)(cg * T * C) - (int)H >= (int)FAST_REPAIR_SHIFT) {
				uint32_t at = 0;
				if (l == 0) at = atomicAdd(&rep[0], 1u);
				at = uni(at);
				if (at < FAST_MAX_REPAIR) {
					if (l == 0) {
						rep[2 + 2 * at] = cg;
						rep[3 + 2 * at] = held_rows;
						atomicOr(&P.pass_flags[FAST_MAX_LEVELS], 1u);
					}
					noted = true;
				}
			}
			if (!noted) zero_acc = 1;
		}
	}
	if (__any(zero_acc) && l == 0) atomicOr(&P.info[v].bail, 1u);

}

/* SCAN: the kernel may meet voices with running-sum phases (it then holds both builds of fast_voice). */
#ifndef FK_MINB
#define FK_MINB 1
#endif
template <int T, int SCAN>
__global__ void __launch_bounds__(1024, FK_MINB) fast_kernel(FastParams P) {
	constexpr int NP = 64 * T;
	constexpr int W = 16;
	extern __shared__ __align__(16) unsigned char lds[];
	const int tid = threadIdx.x;
	const int w = (int)uni((uint32_t)tid >> 6);
	const int l = tid & 63;
	/* a sum pass nobody needs costs a launch, not a table staging */
	if (SCAN == 1 && P.mode != 0 && P.mode <= P.sum_levels && P.pass_flags[P.mode - 1] == 0) return;
	if (SCAN == 1 && P.mode == P.sum_levels + 2 && P.pass_flags[FAST_MAX_LEVELS + 1] == 0) return; /* no chains */
	if (SCAN == 1 && P.only_multi && P.pass_flags[FAST_MAX_LEVELS + 2] == 0) return; /* no voice the single-pass build left out */

	HerpC23 *t23 = (HerpC23 *)lds;
	HerpC01 *t01 = (HerpC01 *)(lds + (size_t)P.n_tabs * WAVE_LEN * sizeof(HerpC23));
	unsigned char *areas = lds + (size_t)P.n_tabs * WAVE_LEN * (sizeof(HerpC23) + sizeof(HerpC01));
	const size_t area_bytes = (size_t)P.n_fast * NP * sizeof(float) + (size_t)P.max_steps * sizeof(unsigned long long);
	float *slots = (float *)(areas + (size_t)w * area_bytes) + l; /* lane's column of every row */
	unsigned long long *carry = (unsigned long long *)(areas + (size_t)w * area_bytes + (size_t)P.n_fast * NP * sizeof(float)); /* per step */
	unsigned long long *lring = nullptr; /* the single-pass build: look-back rings after the waves' areas, zeroed */
	if (SCAN == 2) {
		lring = (unsigned long long *)(areas + (size_t)W * area_bytes);
		lring[tid] = 0;
		static_assert(LOOK_LDS_BYTES == 1024 * sizeof(unsigned long long), "one word per thread");
	}

	for (uint32_t t = 0; t < P.n_tabs; ++t) {
		const uint32_t wave = P.wave_of_tab[t];
		const uint4 *s23 = (const uint4 *)(P.g_c23 + (size_t)wave * WAVE_LEN);
		uint4 *d23 = (uint4 *)(t23 + (size_t)t * WAVE_LEN);
		for (uint32_t i = tid; i < WAVE_LEN; i += 64 * W) d23[i] = s23[i];
		const uint2 *s01 = (const uint2 *)(P.g_c01 + (size_t)wave * WAVE_LEN);
		uint2 *d01 = (uint2 *)(t01 + (size_t)t * WAVE_LEN);
		for (uint32_t i = tid; i < WAVE_LEN; i += 64 * W) d01[i] = s01[i];
	}
	__syncthreads(); /* the only barrier: tables are shared, all else is per wave */

	const uint32_t g = blockIdx.x * W + (uint32_t)w;
	const uint32_t total_waves = gridDim.x * W;
	const uint32_t NV = P.n_voices;
	const uint32_t wpv = total_waves >= NV ? total_waves / NV : 1; /* waves per voice */
	uint32_t v = total_waves >= NV ? g / wpv : g;
	const uint32_t vstride = total_waves >= NV ? NV : total_waves;
	const uint32_t cstart = total_waves >= NV ? g % wpv : 0;

	for (; v < NV; v += vstride) {
		const FastInfo fi = P.info[v];
		const uint32_t seq_kind = SCAN ? uni(fi.seq) : 0u;
		if (SCAN == 1 && P.mode != 0 && P.mode <= P.sum_levels && (seq_kind != 2 || uni(fi.levels) < P.mode))
			continue; /* a sum pass only concerns multi-pass voices that deep */
		if (SCAN == 1 && P.mode == P.sum_levels + 2 && (seq_kind != 2 || uni(fi.n_chain) == 0))
			continue; /* the chain-input pass only concerns voices with feedback chains */
		if (SCAN == 2) { /* the other kinds of running-sum voice have a launch of the full build to themselves */
			if (seq_kind == 1 || seq_kind == 2) continue;
			if (seq_kind == 3) fast_voice<T, 2>(P, v, fi, slots, carry, t23, t01, l, wpv, cstart, total_waves >= NV ? lring : nullptr);
			else fast_voice<T, 0>(P, v, fi, slots, carry, t23, t01, l, wpv, cstart);
			continue;
		}
		if (SCAN == 1 && (P.only_multi ? (seq_kind != 1 && seq_kind != 2) : seq_kind == 3)) continue;
		if (SCAN && seq_kind != 0) fast_voice<T, 1>(P, v, fi, slots, carry, t23, t01, l, wpv, cstart);
		else fast_voice<T, 0>(P, v, fi, slots, carry, t23, t01, l, wpv, cstart);
	}
}

/* The row groups fast_kernel noted (see FAST_REPAIR_SHIFT): same workgroup shape and LDS layout. */
template <int T>
__global__ void __launch_bounds__(1024) repair_kernel(FastParams P) {
	constexpr int NP = 64 * T;
	constexpr int W = 16;
	extern __shared__ __align__(16) unsigned char lds[];
	if (P.pass_flags[FAST_MAX_LEVELS] == 0) return; /* the usual case */
	const int tid = threadIdx.x;
	const int w = (int)uni((uint32_t)tid >> 6);
	const int l = tid & 63;
	HerpC23 *t23 = (HerpC23 *)lds;
	HerpC01 *t01 = (HerpC01 *)(lds + (size_t)P.n_tabs * WAVE_LEN * sizeof(HerpC23));
	unsigned char *areas = lds + (size_t)P.n_tabs * WAVE_LEN * (sizeof(HerpC23) + sizeof(HerpC01));
	const size_t area_bytes = (size_t)P.n_fast * NP * sizeof(float) + (size_t)P.max_steps * sizeof(unsigned long long);
	float *slots = (float *)(areas + (size_t)w * area_bytes) + l;
	unsigned long long *carry = (unsigned long long *)(areas + (size_t)w * area_bytes + (size_t)P.n_fast * NP * sizeof(float));
	for (uint32_t t = 0; t < P.n_tabs; ++t) {
		const uint32_t wave = P.wave_of_tab[t];
		const uint4 *s23 = (const uint4 *)(P.g_c23 + (size_t)wave * WAVE_LEN);
		uint4 *d23 = (uint4 *)(t23 + (size_t)t * WAVE_LEN);
		for (uint32_t i = tid; i < WAVE_LEN; i += 64 * W) d23[i] = s23[i];
		const uint2 *s01 = (const uint2 *)(P.g_c01 + (size_t)wave * WAVE_LEN);
		uint2 *d01 = (uint2 *)(t01 + (size_t)t * WAVE_LEN);
		for (uint32_t i = tid; i < WAVE_LEN; i += 64 * W) d01[i] = s01[i];
	}
	__syncthreads();
	/* one wave per voice with noted groups */
	for (uint32_t v = blockIdx.x * W + (uint32_t)w; v < P.n_voices; v += gridDim.x * W) {
		if (uni(P.repair[(size_t)v * FAST_REPAIR_WORDS]) == 0) continue;
		const FastInfo fi = P.info[v];
		if (uni(fi.total) == 0 || uni(fi.seq) != 0) continue;
		fast_voice<T, 0, true>(P, v, fi, slots, carry, t23, t01, l, 1u, 0u);
	}
}

/* ======================================================================== */
/* feedback chains: lanes = voices                                          */
/* ======================================================================== */
/* The self-modulation recurrence (wosc.h:273-310: feedback -> phase -> table -> sample -> feedback) is one
 * dependent chain per operator, about a hundred nanoseconds per sample whatever the width of the machine.
 * The block loop ran one such chain on one lane of a wave; here a wave runs sixty-four, one per lane, and
 * nothing but the chain. A workgroup is two waves. The CHAIN wave reads its inputs -- base phases and
 * self-modulation amounts, sixteen frames per lane at a time -- from LDS, runs the recurrence and leaves the
 * samples in LDS. The FEEDER wave moves everything else: it fetches the next batch of inputs from the chains'
 * row pairs in HBM (written by the time-parallel passes: fast_voice, chain-input pass), sums phase increments
 * for chains that get those instead of base phases, or -- for chains whose inputs are just their own
 * frequency and amount lines -- evaluates the lines itself, so that such voices need no chain-input pass at
 * all; and it stores the previous batch of samples to the chain's first row, where the final pass takes them
 * (amplitude, mixing into the parent, voice output). One barrier per batch. 4096 chains are 64 workgroups on
 * 64 CUs, and the render takes frames x chain latency. */
constexpr uint32_t CHAIN_BATCH = 16;               /* frames per lane and batch */
constexpr uint32_t CHAIN_IO_WORDS = CHAIN_BATCH * 64; /* one array of one batch */
constexpr size_t CHAIN_IO_BYTES = (size_t)(2 * 2 + 2) * CHAIN_IO_WORDS * 4; /* in[2][2] + out[2] */

/* LDS layout of a batch array: frame 4q + r of lane l at word (q * 64 + l) * 4 + r -- a lane's four 16-byte
 * accesses are conflict-free */
__device__ __forceinline__ uint32_t chain_io_word(uint32_t q, int l) { return (q * 64u + (uint32_t)l) * 4u; }

/* SMALL: every feedback offset of the batch is known to stay below 2^20 cycles in magnitude, where the short
 * rounding form is exact (rint32w_p31_small) -- no per-sample test on the chain; the caller verifies the bound
 * it assumed for |fb_s| afterwards (fb_max) and redoes the batch without SMALL if it was exceeded. */
template <bool LDS_TAB, bool TAIL, bool SMALL>
__device__ __forceinline__ void chain_batch(const uint4 *bq, const float4 *aq, float4 *sq, uint32_t t, uint32_t n,
		uint32_t tab23, uint32_t tab01, const HerpC23 *g23, const HerpC01 *g01, float dscale, float doff,
		uint32_t &prev_phase, double &prev_Is, float &prev_s, float &fb_s, float &fb_max) {
	/* one 16-byte and one 8-byte LDS read per sample (ds_read_b128 / ds_read_b64): the entries are that aligned */
	typedef double __attribute__((ext_vector_type(2))) f64x2;
	typedef float __attribute__((ext_vector_type(2))) f32x2;
	typedef const f64x2 __attribute__((address_space(3))) *lds_f64x2;
	typedef const f32x2 __attribute__((address_space(3))) *lds_f32x2;
#pragma unroll
	for (int u = 0; u < 4; ++u) {
		const uint32_t b4[4] = {bq[u].x, bq[u].y, bq[u].z, bq[u].w};
		const float a4[4] = {aq[u].x, aq[u].y, aq[u].z, aq[u].w};
		float s4[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const float p = fb_s * a4[j];
			uint32_t ofs = rint32w_p31_small(p);
			if (!SMALL) { if (__builtin_expect(!(fabsf(p) < 0x1p20f), 0)) ofs = rint32w(p * 0x1p31f); }
			const uint32_t phase = b4[j] + ofs;
			const int32_t d = (int32_t)(phase - prev_phase);
			const uint32_t ind = phase >> SLEN_BITS;
			HerpC23 hi; HerpC01 lo;
			if (LDS_TAB) {
				const f64x2 c23 = *(lds_f64x2)(uintptr_t)(tab23 + ind * (uint32_t)sizeof(HerpC23));
				const f32x2 c01 = *(lds_f32x2)(uintptr_t)(tab01 + ind * (uint32_t)sizeof(HerpC01));
				hi.c3 = c23.x; hi.c2 = c23.y; lo.c1 = c01.x; lo.c0 = c01.y;
			} else {
				hi = g23[ind]; lo = g01[ind];
			}
			const double Isv = herp_poly(hi, lo, phase);
			const float sv_new = wosc_diff(Isv, prev_Is, d, dscale, doff);
			bool hold = d == 0; /* wosc.h:292-293: a repeated phase holds the previous sample */
			const bool act = !TAIL || t + (uint32_t)(4 * u + j) < n;
			if (TAIL) hold = hold || !act;
			const float sv = hold ? prev_s : sv_new;
			prev_Is = hold ? prev_Is : Isv;
			prev_phase = (TAIL && !act) ? prev_phase : phase; /* (equal to the old one when held) */
			prev_s = sv;
			s4[j] = sv;
			const float fb_n = (fb_s + sv) * 0.5f;
			fb_s = (TAIL && !act) ? fb_s : fb_n;
			if (SMALL) fb_max = fmaxf(fb_max, fabsf(fb_n)) + fb_n * 0.f; /* (beside the chain, not on it; NaN and Inf stick) */
		}
		sq[u] = make_float4(s4[0], s4[1], s4[2], s4[3]);
	}
}

/* Sixteen consecutive values of a line, frames [t, t + 16) of the segment. Lanes hold different lines: the shape
 * is tested once per batch and shape (not once per value), each shape's loop compiled with its type known. */
template <uint32_t TYPE>
__device__ __forceinline__ void line_batch_shape(const FastLine &fl, uint32_t t, float *out) {
	if (fl.sw.type != TYPE) return;
	Sweep sw = fl.sw;
	sw.type = TYPE;
#pragma unroll
	for (uint32_t j = 0; j < CHAIN_BATCH; ++j)
		if (t + j < fl.goal_len) out[j] = sweep_value_inl<true>(sw, t + j);
}
__device__ __forceinline__ void line_batch(const FastLine &fl, uint32_t t, float *out) {
#pragma unroll
	for (uint32_t j = 0; j < CHAIN_BATCH; ++j) out[j] = fl.hold;
	if (t >= fl.goal_len) return;
	line_batch_shape<LN_cos>(fl, t, out); line_batch_shape<LN_lin>(fl, t, out); line_batch_shape<LN_sah>(fl, t, out);
	line_batch_shape<LN_xpe>(fl, t, out); line_batch_shape<LN_lge>(fl, t, out); line_batch_shape<LN_sqe>(fl, t, out);
	line_batch_shape<LN_cub>(fl, t, out); line_batch_shape<LN_smo>(fl, t, out); line_batch_shape<LN_ncl>(fl, t, out);
	line_batch_shape<LN_nhl>(fl, t, out); line_batch_shape<LN_uwh>(fl, t, out);
	/* (LN_exp / LN_log were resolved to xpe / lge when the sweep was set up: sau/line.c:125-148) */
}

/* the feeder's share of one batch: inputs of frames [t, t + 16) into the LDS arrays */
__device__ __forceinline__ void chain_feed(const ChainDesc &cd, bool live, int l, uint32_t t, uint32_t *acc,
		const uint4 *bp, const float4 *ap, uint32_t *in_base, float *in_amt) {
	if (!live) return;
	uint32_t a = *acc, a_end = *acc; /* a_end: the accumulator after the segment's last frame, should it fall in this batch */
	if (cd.mode == CM_INLINE) {
		/* frequency and amounts from the operator's own lines (sau/line.c fills are functions of the position) */
		float fv[CHAIN_BATCH], m[CHAIN_BATCH];
		uint32_t b[CHAIN_BATCH];
		line_batch(cd.pl, t, m);
		if (!(cd.lflags & CL_FCONST)) line_batch(cd.fl, t, fv);
#pragma unroll
		for (uint32_t j = 0; j < CHAIN_BATCH; ++j) {
			const uint32_t i = t + j;
			uint32_t inc = cd.inc_const;
			if (!(cd.lflags & CL_FCONST)) {
				float v = fv[j];
				if (cd.lflags & (i < cd.fl.goal_len ? CL_MUL_GOAL : CL_MUL_HOLD)) v *= cd.mulc;
				const float x = cd.coeff * v;
				inc = fabsf(x) < 0x1p50f ? (uint32_t)__double2loint((double)x + 0x1.8p52) : rint32w(x);
			}
			a += inc; /* wosc.h:145: pre-increment */
			if (i < cd.n) a_end = a;
			b[j] = a;
		}
#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) {
			*(uint4 *)(in_base + chain_io_word(q, l)) = make_uint4(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
			*(float4 *)(in_amt + chain_io_word(q, l)) = make_float4(m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]);
		}
		*acc = a_end;
		return;
	}
#pragma unroll
	for (uint32_t q = 0; q < 4; ++q) {
		uint4 b = bp[t / 4 + q];
		if (cd.mode == CM_INC) { /* phase increments: summed here */
			const uint32_t i = t + 4 * q;
			b.x += a; b.y += b.x; b.z += b.y; b.w += b.z;
			a = b.w;
			a_end = i + 3 < cd.n ? b.w : i + 2 < cd.n ? b.z : i + 1 < cd.n ? b.y : i < cd.n ? b.x : a_end;
		}
		*(uint4 *)(in_base + chain_io_word(q, l)) = b;
		*(float4 *)(in_amt + chain_io_word(q, l)) = ap[t / 4 + q];
	}
	*acc = a_end;
}

__global__ void __launch_bounds__(128) chain_kernel(FastParams P) {
	extern __shared__ __align__(16) unsigned char lds[];
	if (P.pass_flags[FAST_MAX_LEVELS + 1] == 0) return; /* no voice of the segment has a chain */
	const int l = threadIdx.x & 63;
	const bool feeder = uni((uint32_t)threadIdx.x >> 6) != 0;
	const uint32_t c = blockIdx.x * 64 + (uint32_t)l;
	ChainDesc cd;
	memset(&cd, 0, sizeof cd);
	if (c < P.n_chain_rows && P.chain_desc[c].n != 0) cd = P.chain_desc[c]; /* (an unused pair has only `n` set) */
	/* this launch's share of the chain: frames [c_lo, n) of the segment, n cut at the chunk's end */
	const uint32_t c_lo = P.range_mode ? P.f_lo : 0u;
	uint32_t n = cd.n;
	if (P.range_mode && n > P.f_hi) n = P.f_hi;
	if (n <= c_lo) n = 0;
	if (!__any(n != 0)) return;
	HerpC23 *t23 = (HerpC23 *)lds;
	HerpC01 *t01 = (HerpC01 *)(lds + (size_t)P.n_ctabs * WAVE_LEN * sizeof(HerpC23));
	uint32_t *io = (uint32_t *)(lds + (size_t)P.n_ctabs * WAVE_LEN * (sizeof(HerpC23) + sizeof(HerpC01)));
	for (uint32_t t = 0; t < P.n_ctabs; ++t) {
		const uint32_t wave = P.cwave_of_tab[t];
		const uint4 *s23 = (const uint4 *)(P.g_c23 + (size_t)wave * WAVE_LEN);
		uint4 *d23 = (uint4 *)(t23 + (size_t)t * WAVE_LEN);
		for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += 128) d23[i] = s23[i];
		const uint2 *s01 = (const uint2 *)(P.g_c01 + (size_t)wave * WAVE_LEN);
		uint2 *d01 = (uint2 *)(t01 + (size_t)t * WAVE_LEN);
		for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += 128) d01[i] = s01[i];
	}
	const uint32_t wave = cd.wave < 12 ? cd.wave : 0;
	const int ti = P.ctab_of_wave[wave];
	const bool all_lds = __all(n == 0 || ti >= 0) != 0;
	const uint32_t tab23 = (uint32_t)(uintptr_t)(t23 + (size_t)(ti >= 0 ? ti : 0) * WAVE_LEN);
	const uint32_t tab01 = (uint32_t)(uintptr_t)(t01 + (size_t)(ti >= 0 ? ti : 0) * WAVE_LEN);
	const HerpC23 *g23 = P.g_c23 + (size_t)wave * WAVE_LEN;
	const HerpC01 *g01 = P.g_c01 + (size_t)wave * WAVE_LEN;
	const float dscale = P.wc[wave].diff_scale, doff = P.wc[wave].diff_offset;
	DevOp &o = P.ops[cd.gop];
	/* row pair of the chain (idle lanes: pair 0, reads only) */
	float *brow = P.chain_rows + (size_t)2 * (n ? c : 0u) * P.chain_stride;
	const uint4 *bp = (const uint4 *)brow;
	const float4 *ap = (const float4 *)(brow + P.chain_stride);
	float4 *op = (float4 *)brow;
	/* frames count from the chunk's start below: t = c_lo + (batch index) * 16 */
	const uint32_t n_rel = n ? n - c_lo : 0u;
	uint32_t n_all = n ? (n_rel & ~(CHAIN_BATCH - 1)) : 0xfffffff0u, n_max = n_rel;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		n_all = min(n_all, (uint32_t)__shfl_xor((int)n_all, d));
		n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, d));
	}
	n_all = uni(n_all); n_max = uni(n_max);
	/* in[b][0]: base phases, in[b][1]: amounts, then out[b]: samples; b = batch & 1 */
	auto in_base = [&](uint32_t b) { return io + (size_t)(2 * b) * CHAIN_IO_WORDS; };
	auto in_amt = [&](uint32_t b) { return (float *)(io + (size_t)(2 * b + 1) * CHAIN_IO_WORDS); };
	auto out_s = [&](uint32_t b) { return (float *)(io + (size_t)(4 + b) * CHAIN_IO_WORDS); };
	const uint32_t n_batches = (n_max + CHAIN_BATCH - 1) / CHAIN_BATCH;
	if (feeder) {
		uint32_t acc = c_lo ? o.st_phase : o.phase; /* CM_INC, CM_INLINE: the phase accumulator (staged by the chunk before) */
		/* step k: feed batch k while the chain wave runs batch k - 1, store the samples of batch k - 2 */
		for (uint32_t k = 0; k <= n_batches; ++k) {
			if (k < n_batches && c_lo + k * CHAIN_BATCH < n)
				chain_feed(cd, true, l, c_lo + k * CHAIN_BATCH, &acc, bp, ap, in_base(k & 1), in_amt(k & 1));
			if (k >= 2 && c_lo + (k - 2) * CHAIN_BATCH < n) {
				const float *sq = out_s(k & 1);
#pragma unroll
				for (uint32_t q = 0; q < 4; ++q) op[(c_lo + (k - 2) * CHAIN_BATCH) / 4 + q] = *(const float4 *)(sq + chain_io_word(q, l));
			}
			__syncthreads(); /* (the first one also: tables staged) */
		}
		if (n_batches && c_lo + (n_batches - 1) * CHAIN_BATCH < n) {
			const float *sq = out_s((n_batches - 1) & 1);
#pragma unroll
			for (uint32_t q = 0; q < 4; ++q) op[(c_lo + (n_batches - 1) * CHAIN_BATCH) / 4 + q] = *(const float4 *)(sq + chain_io_word(q, l));
		}
		if (n && cd.mode != CM_BASE && !(cd.mode == CM_INLINE && (cd.lflags & CL_FCONST))) o.st_phase = acc;
		return;
	}
	/* ---- the chain wave ---- */
	/* the operator's state, or what the chunk before this one staged */
	uint32_t prev_phase = c_lo ? o.st_prev_phase : o.prev_phase;
	double prev_Is = c_lo ? o.st_prev_Is : o.prev_Is;
	float prev_s = c_lo ? o.st_prev_s : o.prev_s, fb_s = c_lo ? bits_f(o.ras_alpha) : o.fb_s;
	__syncthreads();
	if (n && c_lo == 0 && (o.flags & OPF_OSC_RESET)) { /* wosc.h:215-231 with the first base phase, as the block loop does */
		const uint32_t phase00 = in_base(0)[chain_io_word(0, l)];
		const uint32_t pa = phase00 - SLEN;
		prev_Is = herp_poly(g23[pa >> SLEN_BITS], g01[pa >> SLEN_BITS], pa);
		const double Is0 = herp_poly(g23[phase00 >> SLEN_BITS], g01[phase00 >> SLEN_BITS], phase00);
		prev_s = wosc_diff(Is0, prev_Is, (int32_t)SLEN, dscale, doff);
		prev_Is = Is0;
		prev_phase = phase00;
	}
	for (uint32_t k = 0; k < n_batches; ++k) {
		const uint32_t t = c_lo + k * CHAIN_BATCH;
		uint4 bq[4]; float4 aq[4]; float4 sq[4];
		const uint32_t *ib = in_base(k & 1);
		const float *ia = in_amt(k & 1);
#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) { bq[q] = *(const uint4 *)(ib + chain_io_word(q, l)); aq[q] = *(const float4 *)(ia + chain_io_word(q, l)); }
		/* the short rounding form needs |fb_s * amount| < 2^20: amounts below 2^14 and |fb_s| <= 64 (checked after) */
		float a_max = 0.f;
#pragma unroll
		for (int u = 0; u < 4; ++u)
			a_max = fmaxf(fmaxf(a_max, fmaxf(fabsf(aq[u].x), fabsf(aq[u].y))), fmaxf(fabsf(aq[u].z), fabsf(aq[u].w)));
		const uint32_t s_prev_phase = prev_phase; const double s_prev_Is = prev_Is;
		const float s_prev_s = prev_s, s_fb_s = fb_s;
		float fb_max = fabsf(fb_s);
		bool small = !__any(n != 0 && !(a_max < 0x1p14f));
		const bool tail = !((k + 1) * CHAIN_BATCH <= n_all);
#define SAU_CHAIN_BATCH(L, TL, SM) chain_batch<L, TL, SM>(bq, aq, sq, t, n, tab23, tab01, g23, g01, dscale, doff, prev_phase, prev_Is, prev_s, fb_s, fb_max)
		if (small) {
			if (all_lds) { if (tail) SAU_CHAIN_BATCH(true, true, true); else SAU_CHAIN_BATCH(true, false, true); }
			else { if (tail) SAU_CHAIN_BATCH(false, true, true); else SAU_CHAIN_BATCH(false, false, true); }
			if (__any(n != 0 && !(fb_max <= 64.f))) { /* (never seen: feedback is an average of samples) */
				small = false;
				prev_phase = s_prev_phase; prev_Is = s_prev_Is; prev_s = s_prev_s; fb_s = s_fb_s;
			}
		}
		if (!small) {
			if (all_lds) { if (tail) SAU_CHAIN_BATCH(true, true, false); else SAU_CHAIN_BATCH(true, false, false); }
			else { if (tail) SAU_CHAIN_BATCH(false, true, false); else SAU_CHAIN_BATCH(false, false, false); }
		}
#undef SAU_CHAIN_BATCH
		float *os = out_s(k & 1);
#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) *(float4 *)(os + chain_io_word(q, l)) = sq[q];
		__syncthreads();
	}
	if (n) { /* staged: finalize_kernel makes it the operator's state unless the voice's segment is redone */
		o.st_prev_phase = prev_phase;
		o.st_prev_Is = prev_Is;
		o.st_prev_s = prev_s;
		o.ras_alpha = f_bits(fb_s);
		o.ras_level = CHAIN_MARK;
	}
}

/* Apply the closed forms to the operator state, or hand the whole segment
 * to the block loop when a chunk had to bail out. */
__global__ void __launch_bounds__(64) finalize_kernel(FastParams P) {
	/* one thread per (voice, operator); the voice's own bookkeeping goes to its operator 0 */
	const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
	const uint32_t v = gid / P.max_ops, i = gid % P.max_ops;
	if (gid < FAST_FLAGS && P.pass_flags) P.pass_flags[gid] = 0; /* for the next segment's kernels */
	if (v >= P.n_voices) return;
	const FastInfo fi = P.info[v];
	const VoiceDesc vd = P.voices[v];
	if (fi.total == 0 || fi.bail) {
		if (i == 0) {
			P.fast_done[v] = 0;
			P.worklist[atomicAdd(P.work_count, 1u)] = v;
		}
		return;
	}
	const uint32_t *ids = P.op_ids + vd.ops_ofs;
	const uint32_t total = fi.total;
	if (i < vd.nops) {
		DevOp &o = P.ops[ids[i]];
		if (!o.rt_frozen) { /* (out of time: state stands still) */
		if (!(o.flags & OPF_TIME_INF)) o.time -= total;
		const bool o_osc = o.type == OT_WAVE || o.type == OT_RASEG;
		const Step *plan = P.steps + vd.plan_ofs;
		for (uint32_t ln = 0; ln < L_COUNT; ++ln) {
			/* the lines the reference runs or skips for this operator (generator.c:505-664, 756-762) */
			if (ln == L_PAN && i != vd.carr_local) continue;
			if (!o_osc && (ln == L_FREQ || ln == L_FREQ2 || ln == L_PMA)) continue;
			LineState ls = o.line[ln];
			if (ls.flags & LP_GOAL) {
				/* a range partner without range modulators is skipped, not run (generator.c:468-470) */
				bool skipped = false;
				if (ln == L_FREQ2 || ln == L_AMP2) {
					skipped = true;
					for (uint32_t si = 0; si < vd.plan_len; ++si)
						if (plan[si].op == i && plan[si].kind == ST_LINE && plan[si].which == ln) { skipped = false; break; }
				}
				/* a frequency ramp whose goal and state disagree about being ratios rescales its
				 * state by the parent's frequency (sau/line.c:358-370); such a voice only comes
				 * this way when that frequency is one value (analyze_kernel) */
				bool have_mul = false; float mul0 = 0.f;
				const bool g_ratio = (ls.flags & LP_GOAL_RATIO) != 0, s_ratio = (ls.flags & LP_STATE_RATIO) != 0;
				if (!skipped && (ln == L_FREQ || ln == L_FREQ2) && g_ratio != s_ratio) {
					for (uint32_t si = 0; si < vd.plan_len; ++si) {
						const Step st = plan[si];
						if (st.op != i || st.fmul == NO_SLOT || st.prov == NO_SLOT) continue;
						if ((st.kind == ST_LINE && st.which == ln) || (st.kind == ST_OSC && ln == L_FREQ)) {
							have_mul = true; mul0 = P.ops[ids[st.prov]].rt_fconst;
							break;
						}
					}
				}
				if (skipped) line_skip(ls, total, vd.lat, 0);
				else (void)line_begin(ls, total, have_mul, mul0, vd.lat, 0);
			} else {
				line_advance_hold(ls, total, vd.lat, 0);
			}
			o.line[ln] = ls;
		}
		if (o.type == OT_WAVE) {
			if (o.rt_fconst_valid) o.phase += rint32w(o.coeff * o.rt_fconst) * total;
			else o.phase = o.st_phase; /* running sum, staged by the sequential scan */
			o.prev_phase = o.st_prev_phase;
			o.prev_Is = o.st_prev_Is;
			o.prev_s = o.st_prev_s;
			o.flags &= ~OPF_OSC_RESET;
			if (o.ras_level == CHAIN_MARK) { /* a feedback chain: chain_kernel staged the rest of its state */
				o.fb_s = bits_f(o.ras_alpha);
				o.ras_level = 0;
			}
		} else if (o.type == OT_RASEG) {
			const bool rate2x = (o.flags & OPF_RATE2X) != 0;
			const unsigned long long inc64 = (unsigned long long)rint64((rate2x ? o.coeff * 2 : o.coeff) * o.rt_fconst);
			if (o.rt_fconst_valid) o.cycle_phase += inc64 * total;
			else o.cycle_phase = (unsigned long long)__double_as_longlong(o.st_prev_Is); /* running sum, staged */
		} else if (o.type == OT_NOISE) {
			const uint32_t n0 = o.noise_n;
			if (o.wave == NZ_vi) o.noise_prev = ranfast32(n0 + total - 1);
			else if (o.wave == NZ_bv) o.noise_prev = (uint32_t)noise_bv_term(n0 + total - 1);
			o.noise_n = n0 + total;
		}
		}
	}
	if (i != 0) return;
	P.fast_done[v] = total;
	if (total < vd.run_len) {
		P.worklist[atomicAdd(P.work_count, 1u)] = v;
	} else { /* whole segment done here: tell the mixer */
		VoiceOut vo;
		vo.pan_const = P.ops[ids[vd.carr_local]].line[L_PAN].v0;
		vo.has_pan = vd.pan_dynamic_row != ~0u ? 1u : 0u;
		vo.valid_len = total;
		vo.pan_row = vd.pan_dynamic_row;
		P.vinfo[vd.out_row] = vo;
	}
}

struct MixStream {
	uint32_t first_row, n_rows;
	float amp_scale;
	uint32_t write_len;
	int16_t *pcm; /* stream's PCM row */
};

struct MixParams {
	const MixStream *streams;
	const float *vout;
	const float *pan;
	const VoiceOut *vinfo;
	uint32_t row_stride;
	uint32_t pcm_offset;
	uint32_t stereo;
	uint32_t swap_bytes; /* big-endian PCM for AU files (player/sndfile.c:160-168) */
};

/* generator.c:749-825: ordered voice sum (ref-build association) and PCM.
 * One thread per output frame walks the stream's voices in ascending id --
 * the reference's f32 accumulation order -- so the sum is bit-identical to
 * the CPU's and independent of scheduling.  Loads are issued eight voices
 * ahead of the (serially dependent) adds. */
constexpr int MIX_TILE = 256; /* voices whose constants are staged at a time */
constexpr int MIX_AHEAD = 16; /* loads in flight per thread */
__global__ void __launch_bounds__(256) mix_kernel(MixParams P) {
	__shared__ float s_pan[MIX_TILE];
	__shared__ uint32_t s_valid[MIX_TILE];
	__shared__ uint32_t s_prow[MIX_TILE]; /* pan row, or ~0u */
	__shared__ uint32_t s_special;        /* tile has a short row or a pan row */
	const MixStream ms = P.streams[blockIdx.y];
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (blockIdx.x * 256 >= ms.write_len) return;
	const bool act = i < ms.write_len;
	float L = 0.f, R = 0.f;
	for (uint32_t r0 = 0; r0 < ms.n_rows; r0 += MIX_TILE) {
		const uint32_t nt = min((uint32_t)MIX_TILE, ms.n_rows - r0);
		__syncthreads();
		if (threadIdx.x == 0) s_special = 0;
		__syncthreads();
		if (threadIdx.x < nt) {
			const VoiceOut vo = P.vinfo[ms.first_row + r0 + threadIdx.x];
			s_pan[threadIdx.x] = vo.pan_const;
			s_valid[threadIdx.x] = vo.valid_len;
			s_prow[threadIdx.x] = vo.has_pan ? vo.pan_row : ~0u;
			if (vo.has_pan || vo.valid_len < ms.write_len) s_special = 1;
		}
		__syncthreads();
		if (!act) continue;
		const float *base = P.vout + (size_t)(ms.first_row + r0) * P.row_stride + i;
		if (s_special == 0) {
			/* every row of the tile covers the whole segment with a constant pan */
			uint32_t r = 0;
			for (; r + MIX_AHEAD <= nt; r += MIX_AHEAD) {
				float sv[MIX_AHEAD];
#pragma unroll
				for (int u = 0; u < MIX_AHEAD; ++u) sv[u] = base[(size_t)(r + u) * P.row_stride];
#pragma unroll
				for (int u = 0; u < MIX_AHEAD; ++u) {
					const float v = sv[u] * ms.amp_scale;
					const float s_r = v * s_pan[r + u];
					L = (L + v) - s_r;
					R = (R + v) + s_r;
				}
			}
			for (; r < nt; ++r) {
				const float v = base[(size_t)r * P.row_stride] * ms.amp_scale;
				const float s_r = v * s_pan[r];
				L = (L + v) - s_r;
				R = (R + v) + s_r;
			}
			continue;
		}
		for (uint32_t r = 0; r < nt; r += 8) {
			float sv[8], pn[8];
			bool okv[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const uint32_t rr = r + u;
				const bool ok = rr < nt && i < s_valid[rr < nt ? rr : 0];
				okv[u] = ok;
				sv[u] = ok ? base[(size_t)rr * P.row_stride] : 0.f;
				const uint32_t pr = rr < nt ? s_prow[rr] : ~0u;
				pn[u] = rr < nt ? s_pan[rr] : 0.f;
				if (ok && pr != ~0u) pn[u] = P.pan[(size_t)pr * P.row_stride + i];
			}
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				if (okv[u]) { /* generator.c:842-843: a voice adds only the frames it produced */
					const float v = sv[u] * ms.amp_scale;
					const float s_r = v * pn[u];
					L = (L + v) - s_r;
					R = (R + v) + s_r;
				}
			}
		}
	}
	if (!act) return;
	if (P.stereo) {
		int16_t *d = ms.pcm + 2 * (size_t)(P.pcm_offset + i);
		const int16_t l16 = pcm16(L), r16 = pcm16(R);
		d[0] = P.swap_bytes ? pcm_swap(l16) : l16;
		d[1] = P.swap_bytes ? pcm_swap(r16) : r16;
	} else {
		const int16_t m16 = pcm16((L + R) * 0.5f);
		ms.pcm[P.pcm_offset + i] = P.swap_bytes ? pcm_swap(m16) : m16;
	}
}

__global__ void event_kernel(DevOp *ops, const OpUpdate *recs, uint32_t n, const WaveConst *wc) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	OpUpdate u = recs[i];
	DevOp o = ops[u.op];
	apply_update(o, u, wc);
	ops[u.op] = o;
}

/* Known-answer probe of the shared arithmetic as compiled for the device:
 * one block evaluates a line for `len` samples exactly as ST_LINE does. */
__global__ void kat_line_kernel(LineState st, uint32_t len, const float *mul, float *out,
		LineState *st_out) {
	LineState ls = st;
	LineBlock lb = line_begin(ls, len, mul != nullptr, mul ? mul[0] : 0.f, lattice_none(), 0);
	for (uint32_t j = threadIdx.x; j < len; j += blockDim.x)
		out[j] = line_value(lb, j, mul ? mul[j] : 1.f);
	if (threadIdx.x == 0) *st_out = ls;
}

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
		/* what the pool may keep between generators: 16 GiB of the 288 GB of HBM, 1 GiB page-locked;
		 * SAU_AMD_POOL_MB / SAU_AMD_PINNED_POOL_MB set other caps (0: keep nothing) */
		static const size_t cap_dev = env_mb("SAU_AMD_POOL_MB", (size_t)16 << 10);
		static const size_t cap_pin = env_mb("SAU_AMD_PINNED_POOL_MB", (size_t)1 << 10);
		if (held + bytes > (pinned ? cap_pin : cap_dev)) return false;
		(pinned ? pin_ : dev_[dev & 15]).emplace(bytes, q);
		held += bytes;
		return true;
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
	hipStream_t take(int dev) {
		std::lock_guard<std::mutex> lk(mu_);
		auto &v = free_[dev & 15];
		if (v.empty()) return nullptr;
		hipStream_t s = v.back();
		v.pop_back();
		return s;
	}
	void give(int dev, hipStream_t s) {
		std::lock_guard<std::mutex> lk(mu_);
		auto &v = free_[dev & 15];
		if (v.size() < 8) v.push_back(s); else (void)hipStreamDestroy(s);
	}
private:
	std::mutex mu_;
	std::vector<hipStream_t> free_[16];
};

static void *pool_alloc(bool pinned, size_t &bytes, std::string &err) {
	bytes = BufPool::bucket(bytes);
	void *q = BufPool::get().take(pinned, bytes);
	if (q) return q;
	hipError_t e = pinned ? hipHostMalloc(&q, bytes, hipHostMallocDefault) : hipMalloc(&q, bytes);
	if (e != hipSuccess) { err = std::string(pinned ? "hipHostMalloc: " : "hipMalloc: ") + hipGetErrorString(e); return nullptr; }
	return q;
}
static void pool_free(bool pinned, void *q, size_t bytes) {
	if (!q) return;
	if (!BufPool::get().give(pinned, q, bytes)) { if (pinned) (void)hipHostFree(q); else (void)hipFree(q); }
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

class HipBackendImpl : public HipBackend {
public:
	/* every entry point runs on this backend's device, whatever the host thread's current one is
	 * (another generator on another GPU, torch.cuda.set_device): allocations, the pools (keyed by
	 * the current device) and launches all follow it */
	void use_device() { (void)hipSetDevice(dev_); }

	~HipBackendImpl() override {
		use_device();
		/* the buffers go back to the pool (member destructors): nothing may still be using them */
		if (stream_) (void)hipStreamSynchronize(stream_);
		if (chain_stream_) { (void)hipStreamSynchronize(chain_stream_); StreamPool::get().give(dev_, chain_stream_); }
		for (hipEvent_t e : chain_ev_) (void)hipEventDestroy(e);
		for (int i = 0; i < 2; ++i) if (fetch_ev_[i]) (void)hipEventDestroy(fetch_ev_[i]);
		for (auto &e : events_) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
		if (stream_) StreamPool::get().give(dev_, stream_); /* drained above */
	}

	bool init(const BackendConfig &cfg, std::string &err) override {
		cfg_ = cfg;
		int dev = 0;
		const char *env = getenv("SAU_AMD_DEVICE");
		if (env) dev = atoi(env);
		HIP_OK(hipSetDevice(dev));
		dev_ = dev;
		stream_ = StreamPool::get().take(dev);
		if (!stream_) HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
		static std::mutex prop_mu;
		static size_t dev_lds[16]; /* per device: hipGetDeviceProperties costs a millisecond */
		static int dev_cus[16];
		{
			std::lock_guard<std::mutex> lk(prop_mu);
			if (!dev_lds[dev & 15]) {
				hipDeviceProp_t prop;
				HIP_OK(hipGetDeviceProperties(&prop, dev));
				dev_cus[dev & 15] = prop.multiProcessorCount;
				dev_lds[dev & 15] = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor
				                                                          : prop.sharedMemPerBlock;
			}
			lds_limit_ = dev_lds[dev & 15];
			n_cus_ = dev_cus[dev & 15];
		}
		if (lds_limit_ > 160 * 1024) lds_limit_ = 160 * 1024;
		if (const char *ll = getenv("SAU_AMD_LDS_LIMIT")) lds_limit_ = (size_t)atol(ll);
		const char *wt = getenv("SAU_AMD_GEOMETRY"); /* "4x4" (default) or "8x2" */
		geo_ = (wt && !strcmp(wt, "8x2")) ? 0 : 1; /* default 4 waves x 4 samples per lane */
		debug_ = getenv("SAU_AMD_DEBUG") != nullptr;
		fast_enabled_ = getenv("SAU_AMD_NO_FAST") == nullptr;
		seq_enabled_ = getenv("SAU_AMD_NO_SEQ") == nullptr; /* running-sum phases in the time-parallel kernel */
		chain_enabled_ = getenv("SAU_AMD_NO_CHAIN") == nullptr; /* feedback recurrences with lanes = voices */
		chain_inline_ = getenv("SAU_AMD_CHAIN_INLINE") != nullptr;
		inc_rows_enabled_ = getenv("SAU_AMD_NO_INC_ROWS") == nullptr;
		lookback_enabled_ = getenv("SAU_AMD_NO_LOOKBACK") == nullptr; /* single-pass running sums */
		if (const char *lr = getenv("SAU_AMD_LOOK_ROWS")) look_rows_ = (uint32_t)atoi(lr);
		if (const char *cc = getenv("SAU_AMD_CHAIN_CHUNKS")) { /* pipeline depth of a segment with chains (1: off) */
			const int n = atoi(cc);
			chain_chunks_ = n >= 16 ? 16 : n >= 8 ? 8 : n >= 4 ? 4 : n >= 2 ? 2 : 1;
		}
		two_pass_enabled_ = getenv("SAU_AMD_NO_TWO_PASS") == nullptr; /* ... in two passes where possible */
		/* voices per segment from which feedback voices get sixteen one-wave teams per workgroup
		 * (0: never; 1: always, also without feedback -- tests) */
		multi_min_ = 256;
		if (const char *fr = getenv("SAU_AMD_FAST_ROWS")) { /* 8 (default), 4 or 2 */
			const int r = atoi(fr);
			fast_rows_ = r >= 8 ? 8 : r >= 4 ? 4 : 2;
		}
		if (const char *mm = getenv("SAU_AMD_MULTI_MIN")) multi_min_ = (uint32_t)atol(mm);
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
		if (!pcm_.ensure(pcm_row_ * cfg_.n_streams, err)) return false;
		HIP_OK(hipMemset(pcm_.p, 0, pcm_.cap * sizeof(int16_t)));
		vout_rows_ = 0; /* re-sized on the next render */
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

	bool clear_pcm(uint32_t frames, bool stereo, std::string &err) override {
		use_device();
		(void)frames; (void)stereo;
		if (pcm_.p) HIP_OK(hipMemsetAsync(pcm_.p, 0, pcm_row_ * cfg_.n_streams * sizeof(int16_t), stream_));
		return true;
	}

	template <int W, int T, int V>
	bool launch_render(const RenderParams &rp, uint32_t grid, size_t lds, std::string &err) {
		static size_t configured[16]; /* per device (function attributes are per device) */
		if (lds > configured[dev_ & 15]) {
			HIP_OK(hipFuncSetAttribute((const void *)render_kernel<W, T, V>,
					hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
			configured[dev_ & 15] = lds;
		}
		hipLaunchKernelGGL((render_kernel<W, T, V>), dim3(grid), dim3(64 * W * V), lds, stream_, rp);
		HIP_OK(hipGetLastError());
		return true;
	}

	bool render(const SegmentDesc &seg, std::string &err) override {
		use_device();
		if (!seg.n_voices) return true;
		++acc_launches_; /* segments rendered */
		const size_t tab_bytes = (size_t)WAVE_LEN * (sizeof(HerpC23) + sizeof(HerpC01));
		/* Block-loop geometry. Few voices: W waves share one voice (4x4 or 8x2
		 * frames per lane). Many voices: sixteen single-wave teams per workgroup,
		 * each with its own voice, so that every CU has 16 voices in flight. */
		uint32_t W = geo_ ? 4 : 8, T = geo_ ? 4 : 2, V = 1;
		auto team_size = [&](uint32_t w, uint32_t t) {
			size_t b = (size_t)w * 64 * t * sizeof(float) * seg.n_slots + (size_t)seg.max_ops * sizeof(DevOp) +
				sizeof(Misc) + (size_t)seg.max_steps * sizeof(Step) + 64;
			return (b + 15) & ~(size_t)15;
		};
		if (multi_min_ && seg.n_voices >= multi_min_ && (seg.serial || multi_min_ == 1)) {
			const size_t need_tab = seg.wave_mask ? tab_bytes : 0;
			/* longer blocks amortise the per-step bookkeeping: 255 samples per block measured
			 * 14 % faster than 127 on BASELINE config 5 */
			if (16 * team_size(1, 4) + need_tab <= lds_limit_) { W = 1; T = 4; V = 16; }
			else if (16 * team_size(1, 3) + need_tab <= lds_limit_) { W = 1; T = 3; V = 16; }
			else if (16 * team_size(1, 2) + need_tab <= lds_limit_) { W = 1; T = 2; V = 16; }
			else if (16 * team_size(1, 1) + need_tab <= lds_limit_) { W = 1; T = 1; V = 16; }
		}
		/* a voice with very many block buffers: one wave, one frame per lane (256 B per buffer) */
		if (V == 1 && team_size(W, T) > lds_limit_) { W = 1; T = 1; }
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
		if (!voices_.ensure(seg.n_voices, err) || !vinfo_.ensure(seg.n_voices, err)) return false;
		if (seg.n_voices > vout_rows_ || !vout_.p) {
			HIP_OK(hipStreamSynchronize(stream_));
			if (!vout_.ensure((size_t)seg.n_voices * row_stride_, err)) return false;
			vout_rows_ = seg.n_voices;
		}
		if (seg.n_pan_rows && !pan_.ensure((size_t)seg.n_pan_rows * row_stride_, err)) return false;
		if (!mstreams_.ensure(seg.n_streams, err)) return false;
		/* descriptors go through page-locked staging that is reused once the
		 * previous segment's copies have left it */
		/* in steady state (no event between two segments) the descriptors repeat: upload only changes */
		ms_host_.resize(seg.n_streams);
		uint32_t max_write = 0;
		for (uint32_t s = 0; s < seg.n_streams; ++s) {
			MixStream &m = ms_host_[s];
			memset(&m, 0, sizeof m);
			m.first_row = seg.streams[s].first_voice;
			m.n_rows = seg.streams[s].n_voices;
			m.amp_scale = seg.streams[s].amp_scale;
			m.write_len = seg.streams[s].write_len;
			m.pcm = pcm_.p + pcm_row_ * s;
			if (m.write_len > max_write) max_write = m.write_len;
		}
		const bool same_voices = voices_sent_.size() == seg.n_voices && voices_dev_ == voices_.p &&
			memcmp(voices_sent_.data(), seg.voices, seg.n_voices * sizeof(VoiceDesc)) == 0;
		const bool same_ms = ms_sent_.size() == seg.n_streams && ms_dev_ == mstreams_.p &&
			memcmp(ms_sent_.data(), ms_host_.data(), seg.n_streams * sizeof(MixStream)) == 0;
		if (!same_voices) {
			if (!send(voices_.p, seg.voices, seg.n_voices * sizeof(VoiceDesc), err)) return false;
			voices_sent_.assign(seg.voices, seg.voices + seg.n_voices);
			voices_dev_ = voices_.p;
		}
		if (!same_ms) {
			if (!send(mstreams_.p, ms_host_.data(), seg.n_streams * sizeof(MixStream), err)) return false;
			ms_sent_ = ms_host_;
			ms_dev_ = mstreams_.p;
		}
		rp.voices = voices_.p; rp.steps = steps_.p; rp.op_ids = op_ids_.p; rp.ops = ops_.p;
		rp.vout = vout_.p; rp.pan = pan_.p; rp.vinfo = vinfo_.p;
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
			/* The build with all the running-sum code needs more registers: 8 rows per pass would spill. Where the
			 * single-pass (look-back) build serves, it takes those voices and the closed-form ones at the full rows
			 * per pass, and the full build's launches only see what is left (voices with feedback chains, voices
			 * one wave walks in order). */
			const bool look_split = seq_enabled_ && seg.may_scan && two_pass_enabled_ && lookback_enabled_ && look_rows_ != 0;
			if (seq_enabled_ && seg.may_scan && FT > 4 && !look_split) FT = 4;
			if (look_split && FT > look_rows_) FT = look_rows_ >= 8 ? 8 : look_rows_ >= 4 ? 4 : 2;
			/* block buffers: without frequency blocks, or with them when some voice may need
			 * the sequential scan (ramped or modulated frequencies) */
			const bool seq_ok = seq_enabled_ && seg.may_scan;
			const uint32_t n_fast = seq_ok && seg.n_fast_full > seg.n_fast ? seg.n_fast_full : seg.n_fast;
			auto area_of = [&](uint32_t t) {
				return (size_t)n_fast * 64 * t * sizeof(float) + (size_t)fmax_steps * sizeof(unsigned long long);
			};
			const size_t one_tab = seg.wave_mask ? tab_bytes : 0;
			const size_t look_lds = look_split ? LOOK_LDS_BYTES : 0;
			while (FT > 2 && 16 * area_of(FT) + one_tab + look_lds + 1024 > lds_limit_) FT /= 2;
			{ /* every wave table the segment uses in LDS is worth more than rows per pass (an oscillator whose table
			   * is left out reads it from L2 per sample): fewer rows where that makes them all fit */
				const size_t need = (size_t)__builtin_popcount(seg.wave_mask) * tab_bytes + look_lds + 1024;
				uint32_t t = FT;
				while (t > 4 && 16 * area_of(t) + need > lds_limit_) t /= 2; /* (but not below 4 rows: that costs more) */
				if (16 * area_of(t) + need <= lds_limit_) FT = t;
			}
			const size_t area = area_of(FT);
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
			fp.vout = vout_.p; fp.pan = pan_.p; fp.info = finfo_.p; fp.fast_done = fdone_.p;
			fp.worklist = worklist_.p; fp.work_count = work_count_.p; fp.vinfo = vinfo_.p;
			fp.g_c23 = tables_->c23; fp.g_c01 = tables_->c01; fp.fsteps = (FastStep *)fsteps_.p; fp.flines = (FastLine *)flines_.p; fp.faux = (FastAux *)faux_.p;
			fp.row_stride = row_stride_; fp.n_voices = seg.n_voices; fp.n_fast = n_fast;
			fp.seq_enable = seq_ok ? 1u : 0u; fp.ids_full_ofs = n_steps_total_;
			fp.scan = nullptr; fp.scan_groups = 0; fp.mode = 0;
			if (seq_ok && two_pass_enabled_) {
				/* row groups per voice at most: rows hold at least 32 new frames (H <= 32) */
				fp.scan_groups = seg.len / (32 * FTM) + 2;
				if (!scan_.ensure((size_t)seg.n_voices * FAST_MAX_SCAN * fp.scan_groups, err)) return false;
				fp.scan = scan_.p;
				if (look_split && seg.n_look_rows) {
					/* look-back words: valid for this segment's epoch only, so a fresh block starts out zeroed */
					const unsigned long long *before = look_.p;
					if (!look_.ensure((size_t)seg.n_look_rows * 2 * fp.scan_groups, err)) return false;
					if (look_.p != before || look_epoch_ >= (1u << 30) - 1) {
						HIP_OK(hipMemsetAsync(look_.p, 0, look_.cap * sizeof(unsigned long long), stream_));
						look_epoch_ = 0;
					}
					fp.look = look_.p; fp.look_epoch = ++look_epoch_;
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
			fp.repair_on = getenv("SAU_AMD_NO_REPAIR") ? 0u : 1u;
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
			const bool chains = chain_enabled_ && use_fast && fp.scan && seg.serial && seg.n_chain_rows &&
				seg.len <= sauengine::CHAIN_SEG;
			if (chains) {
				const uint32_t cstride = (seg.len + 63) & ~63u;
				if (!chain_rows_.ensure((size_t)seg.n_chain_rows * 2 * cstride + 64, err) ||
				    !chain_desc_.ensure(seg.n_chain_rows, err) ||
				    !fplines_.ensure((size_t)FAST_LISTS * seg.n_voices * fmax_steps * sizeof(FastLine), err)) return false;
				fp.chain_rows = chain_rows_.p; fp.chain_stride = cstride; fp.n_chain_rows = seg.n_chain_rows;
				fp.chain_desc = chain_desc_.p; fp.fplines = (FastLine *)fplines_.p;
				fp.chain_inline = chain_inline_ ? 1u : 0u;
				uint32_t ct = 0;
				for (int wv = 0; wv < 12; ++wv) {
					fp.ctab_of_wave[wv] = -1;
					if (((seg.wave_mask >> wv) & 1) && (ct + 1) * tab_bytes + CHAIN_IO_BYTES + 1024 <= lds_limit_) {
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
				    16 * area + look_lds + (ft + 1) * tab_bytes + 1024 <= lds_limit_) {
					fp.tab_of_wave[wv] = (int8_t)ft;
					fp.wave_of_tab[ft] = (uint8_t)wv;
					++ft;
				}
			}
			fp.n_tabs = ft;
			TimedPair *ta = timing_on_ ? new_pair(3) : nullptr;
			if (ta) (void)hipEventRecord(ta->a, stream_);
			hipLaunchKernelGGL(analyze_kernel, dim3((seg.n_voices + 63) / 64), dim3(64), 0, stream_, fp);
			if (ta) (void)hipEventRecord(ta->b, stream_);
			if (use_fast) {
				hipLaunchKernelGGL(decode_kernel, dim3(seg.n_voices), dim3(64), 0, stream_, fp);
				const size_t flds = ft * tab_bytes + 16 * area;
				/* build 0: closed-form phases only; 1: every kind of running-sum voice; 2: single-pass voices and closed-form ones */
				const int main_build = !seq_ok ? 0 : look_split ? 2 : 1;
				static const void *const fkernels[3][3] = {
					{(const void *)fast_kernel<2, 0>, (const void *)fast_kernel<4, 0>, (const void *)fast_kernel<8, 0>},
					{(const void *)fast_kernel<2, 1>, (const void *)fast_kernel<4, 1>, (const void *)fast_kernel<8, 1>},
					{(const void *)fast_kernel<2, 2>, (const void *)fast_kernel<4, 2>, (const void *)fast_kernel<8, 2>}};
				static size_t fconfigured[16][3][3];
				auto launch_build = [&](int build, uint32_t rows, uint32_t grid) -> bool {
					const int ri = rows == 8 ? 2 : rows == 4 ? 1 : 0;
					const size_t lds = ft * tab_bytes + 16 * area_of(rows) + (build == 2 ? LOOK_LDS_BYTES : 0);
					size_t &conf = fconfigured[dev_ & 15][build][ri];
					if (lds > conf) {
						HIP_OK(hipFuncSetAttribute(fkernels[build][ri], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
						conf = lds;
					}
					void *args[] = {(void *)&fp};
					HIP_OK(hipLaunchKernel(fkernels[build][ri], dim3(grid), dim3(1024), args, lds, stream_));
					return true;
				};
				/* as many waves per voice as it has row groups (up to 64) when voices are
				 * few, one CU-filling grid at most */
				const uint32_t groups = (seg.len + (60 * FT) - 1) / (60 * FT);
				const unsigned long long want = (unsigned long long)seg.n_voices * (groups < 64 ? groups : 64);
				uint32_t fgrid = (uint32_t)((want + 15) / 16 > FK_GRID ? FK_GRID : (want + 15) / 16);
				if (fgrid > FK_GRID) fgrid = FK_GRID;
				/* look-back waits on other workgroups of the launch: all of them must be resident */
				if (fp.look && n_cus_ > 0 && fgrid > (uint32_t)n_cus_) fgrid = (uint32_t)n_cus_;
				if (fgrid < 1) fgrid = 1;
				TimedPair *tf = timing_on_ ? new_pair(2) : nullptr;
				if (tf) (void)hipEventRecord(tf->a, stream_);
				bool launched = true;
				auto launch_fast = [&](uint32_t mode, uint32_t grid = 0) {
					fp.mode = mode;
					if (main_build == 2) { /* what the single-pass build leaves out: returns at once when there is none */
						fp.only_multi = 1;
						if (!launch_build(1, FTM, grid ? grid : fgrid)) launched = false;
						fp.only_multi = 0;
					} else if (!launch_build(main_build, FT, grid ? grid : fgrid)) {
						launched = false;
					}
				};
				if (main_build == 2) { /* closed-form and single-pass voices: one launch, whole segment */
					fp.mode = fp.sum_levels + 1; fp.only_multi = 0;
					if (!launch_build(2, FT, fgrid)) launched = false;
				}
				if (fp.scan) {
					/* some voice may have running-sum phases: sums per row group, their prefixes, final pass */
					/* (with look-back only voices that have, or may get, feedback chains still take sum passes) */
					for (uint32_t pass = 1; pass <= fp.sum_levels && (!fp.look || seg.n_chain_rows); ++pass) {
						launch_fast(pass);
						hipLaunchKernelGGL(scan_kernel, dim3(seg.n_voices), dim3(64), 0, stream_, fp);
					}
					if (fp.chain_rows) {
						/* The chains' inputs, the chains themselves (lanes = voices), the final pass -- pipelined over
						 * chunks of the segment: chain_kernel occupies one CU per 64 chains for frames x chain latency,
						 * so it runs on a stream of its own while, on the other CUs, the chain-input pass prepares the
						 * chunks after it and the final pass finishes the chunks before it. */
						const size_t clds = (size_t)fp.n_ctabs * tab_bytes + CHAIN_IO_BYTES;
						static size_t cconfigured[16];
						if (clds > cconfigured[dev_ & 15]) {
							HIP_OK(hipFuncSetAttribute((const void *)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
							cconfigured[dev_ & 15] = clds;
						}
						const uint32_t cgrid = (seg.n_chain_rows + 63) / 64;
						uint32_t n_chunks = chain_chunks_;
						while (n_chunks > 1 && seg.len / n_chunks < 4096) n_chunks /= 2;
						if (!chain_stream_ && n_chunks > 1) {
							chain_stream_ = StreamPool::get().take(dev_);
							if (!chain_stream_) HIP_OK(hipStreamCreateWithFlags(&chain_stream_, hipStreamNonBlocking));
						}
						while (chain_ev_.size() < 2 * (size_t)n_chunks) {
							hipEvent_t e;
							HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
							chain_ev_.push_back(e);
						}
						/* chunk boundaries: multiples of the chain kernel's batch */
						const uint32_t clen = ((seg.len + n_chunks - 1) / n_chunks + 255) & ~255u;
						/* the time-parallel passes leave the chains' CUs alone while both run */
						const uint32_t pgrid = n_chunks > 1 && fgrid + cgrid > FK_GRID ? (FK_GRID > cgrid + 32 ? FK_GRID - cgrid : 32) : fgrid;
						TimedPair *tc = timing_on_ ? new_pair(0) : nullptr; /* counted with the block loop it replaces */
						if (n_chunks == 1) {
							fp.range_mode = 0;
							launch_fast(fp.sum_levels + 2);
							if (tc) (void)hipEventRecord(tc->a, stream_);
							hipLaunchKernelGGL(chain_kernel, dim3(cgrid), dim3(128), clds, stream_, fp);
							if (tc) (void)hipEventRecord(tc->b, stream_);
							launch_fast(fp.sum_levels + 1);
						} else {
							for (uint32_t c = 0; c < n_chunks; ++c) { /* inputs of chunk c, then its chains on the other stream */
								fp.range_mode = 1; fp.f_lo = c * clen; fp.f_hi = c + 1 == n_chunks ? 0xffffffffu : (c + 1) * clen;
								fp.range_last = c + 1 == n_chunks;
								launch_fast(fp.sum_levels + 2, pgrid);
								HIP_OK(hipEventRecord(chain_ev_[2 * c], stream_));
								HIP_OK(hipStreamWaitEvent(chain_stream_, chain_ev_[2 * c], 0));
								FastParams cp = fp;
								cp.range_mode = 1; cp.f_lo = c * clen; cp.f_hi = c + 1 == n_chunks ? 0xffffffffu : (c + 1) * clen;
								if (tc && c == 0) (void)hipEventRecord(tc->a, chain_stream_);
								hipLaunchKernelGGL(chain_kernel, dim3(cgrid), dim3(128), clds, chain_stream_, cp);
								if (tc && c + 1 == n_chunks) (void)hipEventRecord(tc->b, chain_stream_);
								HIP_OK(hipEventRecord(chain_ev_[2 * c + 1], chain_stream_));
							}
							for (uint32_t c = 0; c < n_chunks; ++c) { /* the final pass follows the chains chunk by chunk */
								HIP_OK(hipStreamWaitEvent(stream_, chain_ev_[2 * c + 1], 0));
								fp.range_mode = 2; fp.f_lo = c * clen; fp.f_hi = c + 1 == n_chunks ? 0xffffffffu : (c + 1) * clen;
								fp.range_last = c + 1 == n_chunks;
								launch_fast(fp.sum_levels + 1, c + 1 == n_chunks ? fgrid : pgrid);
							}
							fp.range_mode = 0; fp.f_lo = 0; fp.f_hi = 0; fp.range_last = 0;
						}
					} else {
						launch_fast(fp.sum_levels + 1);
					}
				} else {
					launch_fast(0);
				}
				if (!launched) return false;
				{ /* row groups noted for a second evaluation: returns at once when there are none */
					const void *rk = FT == 8 ? (const void *)repair_kernel<8> : FT == 4 ? (const void *)repair_kernel<4>
					                                                                    : (const void *)repair_kernel<2>;
					static size_t rconfigured[16][3];
					size_t &rconf = rconfigured[dev_ & 15][FT == 8 ? 2 : FT == 4 ? 1 : 0];
					if (flds > rconf) {
						HIP_OK(hipFuncSetAttribute(rk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
						rconf = flds;
					}
					const uint32_t rgrid = (seg.n_voices + 15) / 16 < 64 ? (seg.n_voices + 15) / 16 : 64;
					fp.mode = 0;
					if (FT == 8) hipLaunchKernelGGL((repair_kernel<8>), dim3(rgrid), dim3(1024), flds, stream_, fp);
					else if (FT == 4) hipLaunchKernelGGL((repair_kernel<4>), dim3(rgrid), dim3(1024), flds, stream_, fp);
					else hipLaunchKernelGGL((repair_kernel<2>), dim3(rgrid), dim3(1024), flds, stream_, fp);
				}
				if (tf) (void)hipEventRecord(tf->b, stream_);
			}
			TimedPair *tz = timing_on_ ? new_pair(3) : nullptr;
			if (tz) (void)hipEventRecord(tz->a, stream_);
			hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)(((size_t)seg.n_voices * fp.max_ops + 63) / 64)), dim3(64), 0,
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
		TimedPair *tp = timing_on_ ? new_pair(0) : nullptr;
		if (tp) (void)hipEventRecord(tp->a, stream_);
		bool ok = V > 1 ? (T == 4 ? launch_render<1, 4, 16>(rp, block_grid_, lds, err)
		                 : T == 3 ? launch_render<1, 3, 16>(rp, block_grid_, lds, err)
		                 : T == 2 ? launch_render<1, 2, 16>(rp, block_grid_, lds, err)
		                          : launch_render<1, 1, 16>(rp, block_grid_, lds, err))
		        : (W == 1) ? launch_render<1, 1, 1>(rp, block_grid_, lds, err)
		        : geo_ ? launch_render<4, 4, 1>(rp, block_grid_, lds, err)
		               : launch_render<8, 2, 1>(rp, block_grid_, lds, err);
		if (!ok) return false;
		if (tp) (void)hipEventRecord(tp->b, stream_);
		if (debug_) debug_dump("after render", seg);
		if (max_write) {
			MixParams mp;
			mp.streams = mstreams_.p; mp.vout = vout_.p; mp.pan = pan_.p; mp.vinfo = vinfo_.p;
			mp.row_stride = row_stride_; mp.pcm_offset = seg.pcm_offset;
			mp.stereo = seg.stereo ? 1 : 0;
			mp.swap_bytes = seg.swap_bytes ? 1 : 0;
			TimedPair *tm = timing_on_ ? new_pair(1) : nullptr;
			if (tm) (void)hipEventRecord(tm->a, stream_);
			hipLaunchKernelGGL(mix_kernel, dim3((max_write + 255) / 256, seg.n_streams), dim3(256), 0,
					stream_, mp);
			HIP_OK(hipGetLastError());
			if (tm) (void)hipEventRecord(tm->b, stream_);
		}
		return true;
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
		slot &= 1;
		use_device();
		if (!fetch_ev_[slot]) HIP_OK(hipEventCreateWithFlags(&fetch_ev_[slot], hipEventDisableTiming));
		HIP_OK(hipMemcpyAsync(dst, pcm_.p + pcm_row_ * stream,
				(size_t)frames * (stereo ? 2 : 1) * sizeof(int16_t), hipMemcpyDeviceToHost, stream_));
		HIP_OK(hipEventRecord(fetch_ev_[slot], stream_));
		return true;
	}
	bool wait_fetch(int slot, std::string &err) override {
		slot &= 1;
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

	void timing(double *render_ms, double *mix_ms, uint64_t *launches, bool reset) override {
		if (!timing_on_) { timing_on_ = true; }
		(void)hipStreamSynchronize(stream_);
		drain_pairs();
		if (render_ms) *render_ms = acc_ms_[0] + acc_ms_[2];
		if (mix_ms) *mix_ms = acc_ms_[1];
		if (launches) *launches = acc_launches_;
		if (reset) { acc_ms_[0] = acc_ms_[1] = acc_ms_[2] = acc_ms_[3] = 0; acc_launches_ = 0; }
	}

	void *stream_handle() override { return (void *)stream_; }
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
			(void)hipMemcpy(vi.data(), vinfo_.p, vi.size() * sizeof(VoiceOut), hipMemcpyDeviceToHost);
			(void)hipMemcpy(fd.data(), fdone_.p, fd.size() * 4, hipMemcpyDeviceToHost);
			(void)hipMemcpy(fi.data(), finfo_.p, fi.size() * sizeof(FastInfo), hipMemcpyDeviceToHost);
			float v8[8] = {0};
			(void)hipMemcpy(v8, vout_.p, sizeof v8, hipMemcpyDeviceToHost);
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
	uint32_t vout_rows_ = 0;
	WaveConst wconst_[12];
	DevBuf<DevOp> ops_;
	DevBuf<Step> steps_;
	DevBuf<FastIds> fast_ids_; /* [2][n_steps_total_]: without / with frequency blocks */
	uint32_t n_steps_total_ = 0;
	DevBuf<uint32_t> op_ids_;
	DevBuf<VoiceDesc> voices_;
	DevBuf<float> vout_, pan_;
	DevBuf<VoiceOut> vinfo_;
	DevBuf<int16_t> pcm_;
	DevBuf<OpUpdate> recs_;
	DevBuf<MixStream> mstreams_;
	const TableSet *tables_ = nullptr;
	PinBuf<int16_t> h_pcm_; /* fetch_pcm() staging */
	PinBuf<unsigned char> arena_; /* stage() */
	size_t arena_used_ = 0;
	std::vector<VoiceDesc> voices_sent_;   /* what the device copies hold */
	std::vector<MixStream> ms_sent_, ms_host_;
	const void *voices_dev_ = nullptr, *ms_dev_ = nullptr;
	std::deque<TimedPair> events_; /* (a deque: pairs handed out stay where they are while more are added) */
	size_t n_used_ = 0;
	bool timing_on_ = false;
	double acc_ms_[4] = {0, 0, 0, 0};
	uint64_t acc_launches_ = 0;
	bool fast_enabled_ = true;
	int timing_level_ = 2;
	DevBuf<FastInfo> finfo_;
	DevBuf<uint32_t> fdone_, worklist_, work_count_;
	DevBuf<unsigned char> fsteps_, flines_, faux_;
	DevBuf<unsigned long long> scan_;
	DevBuf<uint32_t> pass_flags_, repair_;
	hipEvent_t fetch_ev_[2] = {nullptr, nullptr};
	int dev_ = 0;
	std::map<void *, size_t> host_blocks_; /* alloc_host() blocks and their pool sizes */
	uint32_t multi_min_ = 256;
	uint32_t fast_rows_ = 8;
	bool seq_enabled_ = true, two_pass_enabled_ = true, chain_enabled_ = true, chain_inline_ = false;
	uint32_t chain_chunks_ = 8;
	bool inc_rows_enabled_ = true;
	bool lookback_enabled_ = true;
	uint32_t look_rows_ = 8; /* rows per pass of the single-pass build (SAU_AMD_LOOK_ROWS; 0: no such build, the full one takes every voice) */
	DevBuf<unsigned long long> look_;
	uint32_t look_epoch_ = 0;
	int n_cus_ = 0;
	DevBuf<uint32_t> inc_rows_;
	hipStream_t chain_stream_ = nullptr;
	std::vector<hipEvent_t> chain_ev_;
	DevBuf<float> chain_rows_;
	DevBuf<ChainDesc> chain_desc_;
	DevBuf<unsigned char> fplines_;
	uint32_t block_grid_ = 1;
};

/* Known-answer probe: div_diff_scale(a, b) against IEEE a / b for every f32 b
 * with 1 <= |b| <= 2^31 (a superset of the rounded integers the differentiator
 * divides by). Thread t takes the bit patterns t, t + stride, ... */
__global__ void kat_div_kernel(float a, int variant, unsigned long long *mismatches, uint32_t *first_bad) {
	const uint32_t lo = 0x3f800000u, hi = 0x4f000000u; /* 1.0f .. 2^31 */
	unsigned long long bad = 0;
	for (uint32_t bits = lo + blockIdx.x * blockDim.x + threadIdx.x; bits <= hi; bits += gridDim.x * blockDim.x) {
		for (int sgn = 0; sgn < 2; ++sgn) {
			const float b = bits_f(bits | (sgn ? 0x80000000u : 0u));
			const float want = __fdiv_rn(a, b);
			float got;
			if (variant == 0) {
				got = div_diff_scale(a, b);
			} else { /* no correction: the probe must be able to see this fail */
				got = a * __builtin_amdgcn_rcpf(b);
			}
			if (f_bits(want) != f_bits(got)) { ++bad; atomicMin(first_bad, bits); }
		}
		if (bits > hi - gridDim.x * blockDim.x) break; /* no wrap past the last pattern */
	}
	if (bad) atomicAdd(mismatches, bad);
}

bool kat_div(float a, int variant, unsigned long long *mismatches, uint32_t *first_bad) {
	unsigned long long *d_m = nullptr;
	uint32_t *d_f = nullptr;
	bool ok = hipMalloc((void **)&d_m, sizeof *d_m) == hipSuccess && hipMalloc((void **)&d_f, sizeof *d_f) == hipSuccess;
	if (ok) {
		const uint32_t none = 0xffffffffu;
		ok = hipMemset(d_m, 0, sizeof *d_m) == hipSuccess &&
			hipMemcpy(d_f, &none, sizeof none, hipMemcpyHostToDevice) == hipSuccess;
	}
	if (ok) {
		hipLaunchKernelGGL(kat_div_kernel, dim3(4096), dim3(256), 0, 0, a, variant, d_m, d_f);
		ok = hipDeviceSynchronize() == hipSuccess &&
			hipMemcpy(mismatches, d_m, sizeof *d_m, hipMemcpyDeviceToHost) == hipSuccess &&
			hipMemcpy(first_bad, d_f, sizeof *d_f, hipMemcpyDeviceToHost) == hipSuccess;
	}
	if (d_m) (void)hipFree(d_m);
	if (d_f) (void)hipFree(d_f);
	return ok;
}

bool kat_line(const LineState &st, uint32_t len, const float *mul, float *out, LineState *st_out) {
	float *d_mul = nullptr, *d_out = nullptr;
	LineState *d_st = nullptr;
	bool ok = hipMalloc((void **)&d_out, (len + 1) * sizeof(float)) == hipSuccess &&
		hipMalloc((void **)&d_st, sizeof(LineState)) == hipSuccess;
	if (ok && mul) {
		ok = hipMalloc((void **)&d_mul, (len + 1) * sizeof(float)) == hipSuccess &&
			hipMemcpy(d_mul, mul, len * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
	}
	if (ok) {
		hipLaunchKernelGGL(kat_line_kernel, dim3(1), dim3(256), 0, 0, st, len, d_mul, d_out, d_st);
		ok = hipDeviceSynchronize() == hipSuccess &&
			hipMemcpy(out, d_out, len * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess &&
			hipMemcpy(st_out, d_st, sizeof(LineState), hipMemcpyDeviceToHost) == hipSuccess;
	}
	if (d_mul) (void)hipFree(d_mul);
	if (d_out) (void)hipFree(d_out);
	if (d_st) (void)hipFree(d_st);
	return ok;
}

HipBackend *create_hip_backend(std::string &err) {
	if (device_count() <= 0) {
		err = "no HIP device available (this backend has no CPU fallback)";
		return nullptr;
	}
	return new HipBackendImpl();
}

} /* namespace sauhip */

