/* k_common.h -- part of hip_backend.hip (included there, inside namespace sauhip; not a header of its own).
 * Parameter blocks of the block loop and the wave-level helpers every kernel uses: uniform (scalar) moves,
 * DPP scans, exact roundings, the look-back words of the single-pass running sums. */

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
	float *big_slots;      /* render_kernel<.., HB>: [workgroup][big_stride] floats in HBM: n_slots x 64 block buffers, max_ops operator
	                        * records, max_steps steps */
	uint32_t big_stride;
	int8_t tab_of_wave[12];/* LDS table index per wave id, or -1 */
	uint8_t wave_of_tab[12];
	WaveConst wc[12];
};

struct Misc {
	WaveConst wc[12];      /* per-wave constants, copied from the launch parameters */
	int32_t tab_of_wave[12];
	uint16_t len_stack[MAX_NEST + 1]; /* block lengths per nesting level (<= 1024 each) */
	uint16_t rem_stack[MAX_NEST + 1]; /* frames until the level's operator stops, TAIL_FAR at most (TailCtx.rem) */
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

#include "k_wave_scan.h" /* wave_incl_scan_dpp, wave_incl_scan64_dpp, wave_sum64_dpp, readlane64 */

/* Decoupled look-back over a voice's row groups (FastParams.look). The calling wave owns group cg and its
 * group total `tot`; returns the sum of all earlier groups' totals. Words carry value and status together,
 * so one relaxed device-scope store publishes and one load observes -- no fences. A wave only ever waits
 * for groups before its own: those belong to waves of this launch that are resident (the grid is at most
 * one workgroup per CU) or to an earlier launch of the same segment. */
constexpr uint32_t LOOK_AGG = 1, LOOK_PREFIX = 2;
/* Waiting across workgroups (words in HBM) assumes that the workgroups a wave waits for are resident or next in
 * line for a CU. That holds for one such launch at a time on a device this process has to itself; a second
 * process on the same device (or a CU-masked one) can break it. The wait is therefore bounded: after
 * LOOK_SPIN_MAX empty polls (tens of milliseconds; a normal wait is microseconds) the wave gives up, says so
 * through `stuck`, and the voice's segment is redone by the block loop (FastInfo.bail) -- slow, never hung. */
constexpr uint32_t LOOK_SPIN_MAX = 1u << 15;
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
/* LDS: entry of group i at ent[i % ring] (the caller keeps cg % ring), tag i + 1; HBM: at ent[i], tag = the segment's epoch */
template <bool LDS> __device__ __forceinline__ uint32_t lookback32(unsigned long long *ent, const uint32_t cg, const uint32_t tot,
		const uint32_t epoch, const uint32_t ring, const uint32_t cgm /* LDS: cg % ring */, const int l, uint32_t &stuck,
		const bool withhold = false /* test aid (SAU_AMD_LOOK_WITHHOLD): group 1 never publishes, what waits for it gives up */) {
	/* (LDS: groups up to `ring` back have their entries; what lies further back is dead and reads as empty) */
	auto at = [&](uint32_t i) { const int s_ = (int)cgm - (int)(cg - i); return LDS ? (uint32_t)(s_ < 0 ? s_ + (int)ring : s_) : i; };
	auto tag = [&](uint32_t i) { return LDS ? i + 1 : epoch; };
	if (cg == 0) {
		if (l == 0) look_store<LDS>(&ent[0], look_word(tag(0), LOOK_PREFIX, tot));
		return 0;
	}
	const bool mute = !LDS && withhold && cg == 1;
	if (l == 0 && !mute) look_store<LDS>(&ent[at(cg)], look_word(tag(cg), LOOK_AGG, tot));
	uint32_t excl = 0;
	uint32_t empty = 0; /* polls in a row that found nothing new (words in HBM: bounded, see LOOK_SPIN_MAX) */
	int p = (int)cg - 1; /* the nearest group not yet accounted for */
	for (;;) {
		const int idx = p - l; /* lane l looks at the group l before it; before group 0 the prefix is 0 */
		unsigned long long e = 0;
		if (idx >= 0 && (!LDS || cg - (uint32_t)idx <= ring)) e = look_load<LDS>(&ent[at((uint32_t)idx)]);
		const uint32_t hi = (uint32_t)(e >> 32);
		const uint32_t st = idx < 0 ? LOOK_PREFIX : (hi >> 2) == tag((uint32_t)idx) ? (hi & 3u) : 0u;
		const unsigned long long m_pref = __ballot(st == LOOK_PREFIX), m_none = __ballot(st == 0);
		const int first_pref = m_pref ? __builtin_ctzll(m_pref) : 64;
		const int first_none = m_none ? __builtin_ctzll(m_none) : 64;
		const int upto = first_pref < first_none ? first_pref + 1 : first_none; /* lanes [0, upto) count */
		if (upto) { /* (an empty poll -- nothing new published -- adds nothing: no scan for it) */
			const uint32_t part = wave_incl_scan_dpp(l < upto && idx >= 0 ? (uint32_t)e : 0u);
			excl += (uint32_t)__builtin_amdgcn_readlane((int)part, 63);
		}
		if (first_pref < first_none) break;
		p -= upto;
		if (upto == 0) {
			if (!LDS && ++empty >= LOOK_SPIN_MAX) { stuck = 1; break; } /* give up: the voice's segment goes to the block loop */
			__builtin_amdgcn_s_sleep(LDS ? 1 : 2);
		} else {
			empty = 0;
		}
	}
	/* (after giving up the word still goes out, as a prefix, so that the groups behind do not wait in turn: their sums
	 * are as void as this one's, and the voice is redone) */
	if (l == 0 && !mute) look_store<LDS>(&ent[at(cg)], look_word(tag(cg), LOOK_PREFIX, excl + tot));
	return excl;
}
/* 64-bit totals (R oscillators' cycle counters): low and high halves in two arrays, a pair counts once both
 * words show the same status */
template <bool LDS> __device__ __forceinline__ unsigned long long lookback64(unsigned long long *ent_lo, unsigned long long *ent_hi,
		const uint32_t cg, const unsigned long long tot, const uint32_t epoch, const uint32_t ring, const uint32_t cgm, const int l,
		uint32_t &stuck, const bool withhold = false) {
	auto at = [&](uint32_t i) { const int s_ = (int)cgm - (int)(cg - i); return LDS ? (uint32_t)(s_ < 0 ? s_ + (int)ring : s_) : i; };
	auto tag = [&](uint32_t i) { return LDS ? i + 1 : epoch; };
	auto publish = [&](uint32_t i, uint32_t status, unsigned long long v) {
		look_store<LDS>(&ent_lo[at(i)], look_word(tag(i), status, (uint32_t)v));
		look_store<LDS>(&ent_hi[at(i)], look_word(tag(i), status, (uint32_t)(v >> 32)));
	};
	if (cg == 0) {
		if (l == 0) publish(0, LOOK_PREFIX, tot);
		return 0;
	}
	const bool mute = !LDS && withhold && cg == 1;
	if (l == 0 && !mute) publish(cg, LOOK_AGG, tot);
	unsigned long long excl = 0;
	uint32_t empty = 0;
	int p = (int)cg - 1;
	for (;;) {
		const int idx = p - l;
		unsigned long long a = 0, b = 0;
		if (idx >= 0 && (!LDS || cg - (uint32_t)idx <= ring)) {
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
		if (upto) excl += wave_sum64_dpp(l < upto && idx >= 0 ? v : 0ull);
		if (first_pref < first_none) break;
		p -= upto;
		if (upto == 0) {
			if (!LDS && ++empty >= LOOK_SPIN_MAX) { stuck = 1; break; }
			__builtin_amdgcn_s_sleep(LDS ? 1 : 2);
		} else {
			empty = 0;
		}
	}
	if (l == 0 && !mute) publish(cg, LOOK_PREFIX, excl + tot);
	return excl;
}

/* rint(p * 2^31) wrapped to 32 bits for |p| < 2^20: in f64, p + 1.5 * 2^21
 * has an ulp of 2^-31, so the addition rounds p to a multiple of 2^-31
 * (nearest-even, as llrintf does in the default mode) and leaves that
 * multiple, mod 2^32, in the low word of the significand. */
__device__ __forceinline__ uint32_t rint32w_p31_small(float p) {
	return (uint32_t)__double2loint((double)p + 0x1.8p21);
}
