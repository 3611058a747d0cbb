/* k_fast_voice.h -- part of hip_backend.hip (included there, inside namespace sauhip; not a header of its own).
 * fast_voice / fast_kernel / repair_kernel: rows of 64 frames per wave, closed-form and running-sum phases. */
/* Where entry i of a wave table sits in LDS. Identity: XOR swizzles of the low index bits with the next ones
 * (-DFK_SWZ_BITS=2, 4, 5, tried in r02 against the 58 % bank-conflict cycles of the table gather) left
 * SQ_LDS_BANK_CONFLICT where it was or raised it by 3-5 % and cost 1-5 % in time for the two extra VALU
 * instructions per lookup (profiles/r02_b_lds_swizzle.json): the conflicts are the gather's own. */
/* The voice rows are written once and read once, by another kernel (the mixer), 1.8 GB per config-3 step: streaming
 * (non-temporal) stores here and loads there, so that they do not push anything else out of L2 and the Infinity Cache --
 * 2.51 -> 2.41 ms per 441000-frame step (r03; the loads alone: 2.48). -DFK_TEMPORAL_ROWS: ordinary accesses. */
#ifndef FK_TEMPORAL_ROWS
#define FK_VSTORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define FK_VSTORE(p, v) (*(p) = (v))
#endif
/* ... and so are the chains' rows (inputs written by the chain-input pass, samples read by the final pass: 14 GB per
 * config-5 step; 52.6 -> 51.3 ms, the feeder wave's loads compete with less) */
#ifndef FK_TEMPORAL_ROWS
#define FK_CSTORE(p, v) __builtin_nontemporal_store((v), (p))
#define FK_CLOAD(p) __builtin_nontemporal_load(p)
#else
#define FK_CSTORE(p, v) (*(p) = (v))
#define FK_CLOAD(p) (*(p))
#endif

/* lane l receives lane l-1's value (lane 0: zero; it is lead-in) */
__device__ __forceinline__ uint32_t lane_prev(uint32_t x) {
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ double lane_prev(double x) {
	const uint32_t lo = lane_prev((uint32_t)__double2loint(x));
	const uint32_t hi = lane_prev((uint32_t)__double2hiint(x));
	return __hiloint2double((int)hi, (int)lo);
}

/* The Hermite value from a table block in LDS (FAST_TAB_BYTES: see k_fast_types.h). `tab`: the block's LDS address. */
typedef double __attribute__((ext_vector_type(2))) fk_f64x2;
typedef float __attribute__((ext_vector_type(2))) fk_f32x2;
typedef const fk_f64x2 __attribute__((address_space(3))) *fk_lds_f64x2;
typedef const fk_f32x2 __attribute__((address_space(3))) *fk_lds_f32x2;
struct FkHerp { double c3, c2, c1, c0; };
template <bool WIDE, bool TAB0 = false>
__device__ __forceinline__ FkHerp fk_entry(const uint32_t tab, const uint32_t ph) {
	FkHerp h;
	if (WIDE && TAB0) {
		/* the table block at LDS address 0 (a launch's first table: every operator of a one-wave bank): the address is the
		 * index alone, (ph >> 17) & 0x7ff0 -- two 2-cycle instructions where v_lshrrev + v_lshl_add (a 4-cycle one) stand --
		 * and the [c1, c0] entry's 32 KiB go into the read's offset field. Round 5, measured and not used: a second copy of
		 * the Hermite block under a uniform test of the table's address cost fast_kernel<12, 0, false, true> ten spilled VGPRs
		 * and everything the cheaper address bought (1.784 against 1.757 ms per launch, profiles/r05_headline_ab.json) */
		const uint32_t a = (ph >> (SLEN_BITS - 4)) & ((WAVE_LEN - 1) << 4);
		const fk_f64x2 hi = *(fk_lds_f64x2)(uintptr_t)a;
		const fk_f64x2 lo = *(fk_lds_f64x2)(uintptr_t)(a + FkTab<true>::C01);
		h.c3 = hi.x; h.c2 = hi.y; h.c1 = lo.x; h.c0 = lo.y;
	} else if (WIDE) {
		/* one address (v_lshrrev + v_lshl_add) serves both reads: the [c1, c0] entry sits a constant 32 KiB further on */
		/* (the empty asm keeps LLVM from rewriting (ph >> 21) << 4 as (ph >> 17) & 0x7ff0, which costs a third instruction) */
		uint32_t ind = ph >> SLEN_BITS;
		asm("" : "+v"(ind));
		const uint32_t a = tab + (ind << 4);
		const fk_f64x2 hi = *(fk_lds_f64x2)(uintptr_t)a;
		const fk_f64x2 lo = *(fk_lds_f64x2)(uintptr_t)(a + FkTab<true>::C01);
		h.c3 = hi.x; h.c2 = hi.y; h.c1 = lo.x; h.c0 = lo.y;
	} else {
		const uint32_t ind = ph >> SLEN_BITS;
		const fk_f64x2 hi = *(fk_lds_f64x2)(uintptr_t)(tab + ind * 16u);
		const fk_f32x2 lo = *(fk_lds_f32x2)(uintptr_t)(tab + FkTab<false>::C01 + ind * 8u);
		h.c3 = hi.x; h.c2 = hi.y; h.c1 = (double)lo.x; h.c0 = (double)lo.y;
	}
	return h;
}
/* sau_dev_math.h: herp_poly / herp_poly_rise on such an entry (the same operations in the same order) */
__device__ __forceinline__ double fk_poly(const FkHerp &h, const uint32_t ph) {
	const double x = (double)(ph & (SLEN - 1));
	return ((h.c3 * x + h.c2) * x + h.c1) * x + h.c0;
}
__device__ __forceinline__ double fk_poly_rise(const FkHerp &h, const uint32_t ph) {
	const double x = (double)(ph & (SLEN - 1));
	return ((h.c3 * x + h.c2) * x + h.c1) * x;
}
/* stage the tables a launch uses into its LDS blocks (all threads of the workgroup; a barrier follows) */
template <bool WIDE>
__device__ __forceinline__ void fk_stage_tables(const FastParams &P, unsigned char *lds, const uint32_t tid, const uint32_t nthreads) {
	for (uint32_t t = 0; t < P.n_tabs; ++t) {
		const uint32_t wave = P.wave_of_tab[t];
		const uint4 *s23 = (const uint4 *)(P.g_c23 + (size_t)wave * WAVE_LEN);
		uint4 *d23 = (uint4 *)(lds + (size_t)t * FkTab<WIDE>::BYTES);
		for (uint32_t i = tid; i < WAVE_LEN; i += nthreads) d23[i] = s23[i];
		const HerpC01 *s01 = P.g_c01 + (size_t)wave * WAVE_LEN;
		if (WIDE) {
			fk_f64x2 *d01 = (fk_f64x2 *)(lds + (size_t)t * FkTab<WIDE>::BYTES + FkTab<WIDE>::C01);
			for (uint32_t i = tid; i < WAVE_LEN; i += nthreads) { fk_f64x2 v; v.x = (double)s01[i].c1; v.y = (double)s01[i].c0; d01[i] = v; }
		} else {
			uint2 *d01 = (uint2 *)(lds + (size_t)t * FkTab<WIDE>::BYTES + FkTab<WIDE>::C01);
			for (uint32_t i = tid; i < WAVE_LEN; i += nthreads) d01[i] = ((const uint2 *)s01)[i];
		}
	}
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
template <int T, int SCAN, bool REPAIR = false, bool CUB = false, bool WIDE = false, int SPLIT = 0 /* 1: three loops, 2: the groups between only */, bool TAIL = false>
__device__ __forceinline__ void fast_voice(const FastParams &P, const uint32_t v, const FastInfo &fi,
		float *slots, unsigned long long *carry, const uint32_t tabs /* LDS address of the launch's table blocks */, const int l,
		const uint32_t wpv, const uint32_t cstart, unsigned long long *lring = nullptr,
		const uint32_t dyn_c = 0, const uint32_t dyn_k = 0 /* > 0: chunk dyn_c of dyn_k of the voice's row groups */,
		const uint32_t vpos = ~0u /* the voice's place among the launch's voices when that is not v (look-back lists) */,
		const uint32_t wg_waves = 16u /* waves of a workgroup that render look-back voices (duo_kernel: 8) */) {
	constexpr int NP = 64 * T;
	(void)NP;
	const uint32_t fast_total = uni(fi.total);
	if (fast_total == 0) return;
	if ((uni(fi.cub) != 0) != CUB) return; /* (voices with the loop tails of `cub` R segments: the build with that code, and only it) */
	const VoiceDesc vd = P.voices[v];
	const uint32_t H = uni(fi.H);

	/* this pass's own list of the voice's decoded steps */
	/* SCAN: 0 closed-form phases only; 1 every kind of running-sum voice; 2 single-pass (look-back) voices only,
	 * without the code of the several-pass forms and the feedback chains */
	constexpr bool FULL = SCAN == 1;
	/* SCAN 3: voices with feedback chains and no running sum to scan -- the chain-input and final passes (their own
	 * step lists, launches in chunks of frames, chains' rows) without the several-pass sums */
	constexpr bool CH = SCAN == 1 || SCAN == 3;
	const uint32_t li = CH ? fast_list_of(P.mode, P.sum_levels) : 0u;
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
	const uint32_t w0 = ((vpos != ~0u ? vpos : v) * wpv) % wg_waves; /* the voice's first wave within its workgroup (fast_kernel: waves v * wpv ...) */
	const bool look_lds = SCAN == 2 && lring && wpv >= 2 && w0 + wpv <= wg_waves && !(P.look_wpv_flags & 1u);
	const uint32_t lk_ring = 4 * wpv;
	unsigned long long *lk_base = look_lds ? lring + w0 * 4 : nullptr;
	if (seq && cstart != 0) return;
	const uint32_t gstride = seq ? 1u : wpv;
	unsigned long long *scan = two ? P.scan + (size_t)v * FAST_MAX_SCAN * P.scan_groups : nullptr;
	float *vrow = P.vout + (size_t)vd.out_row * P.row_stride;
	float *prow = (vd.pan_dynamic_row != ~0u) ? P.pan + (size_t)vd.pan_dynamic_row * P.row_stride : nullptr;
	/* A wave renders T rows at a time; a row is 64 consecutive frames, one per lane. Running-sum builds: the first H lanes
	 * of every row are lead-in (recomputed, not stored), rows lie C = 64 - H frames apart. Closed-form builds (round 4,
	 * CONTIG): a group's rows are contiguous in time -- row k + 1 goes on where row k ends, and what its lane 0 needs of
	 * the sample before (phase, Hermite value: prev32 / prev64 below) is row k's lane 63 -- so only a group's first row
	 * pays the lead-in: 64 T - H new frames per group instead of T (64 - H), 5.8 % more at T = 8, H = 4. */
	constexpr bool CONTIG = SCAN == 0 || SCAN == 2; /* (round 4, later: the look-back build too -- its rows chain through sums either way) */
	const uint32_t C = 64u - H;                                   /* new frames of a row that has lead-in lanes */
	const uint32_t RS = CONTIG ? 64u : C;                         /* frames from one row of a group to the next */
	const uint32_t GF = CONTIG ? 64u * T - H : (uint32_t)T * C;   /* new frames per group */
	const uint32_t ngroups = (fast_total + GF - 1) / GF;
	const uint32_t last_group = (fast_total - 1) / GF; /* holds the segment's last frame */
	uint32_t zero_acc = 0; /* nonzero: some hold-previous run could not be resolved here */

	uint32_t *const rep = P.repair + (size_t)v * FAST_REPAIR_WORDS;
	/* REPAIR: the noted row groups instead of all, each evaluated FAST_REPAIR_SHIFT frames early */
	uint32_t n_iter = REPAIR ? min(uni(rep[0]), FAST_MAX_REPAIR) : ngroups;
	uint32_t it_lo = 0;
	if (!REPAIR && SCAN != 2 && P.range_mode != 0) { /* (the single-pass build never runs in chunks) */
		if (seq) { if (!P.range_last) return; } /* one wave in order, carries in LDS: in the last chunk's launch, all of it */
		else {
			const uint32_t tc = GF;
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
	if (!REPAIR && SCAN == 0 && dyn_k) { /* this task's run of consecutive groups */
		(void)fk_chunk_groups(ngroups, dyn_k, P.dyn_small, dyn_c, it_lo, n_iter);
	}
	if constexpr (!REPAIR && SCAN == 0 && WIDE && T == 12 && SPLIT == 0) {
		/* the launch ahead of the INNER build's (fast_kernel): two tasks a voice, its first row group and its last */
		if (P.edge_only) {
			if (dyn_c && last_group == 0) return;
			it_lo = dyn_c ? last_group : 0u;
			n_iter = it_lo + 1;
		}
	}
	/* this voice's launch mixes its stream as it stores the row (FastInfo.tail, k_fast_types.h): rows of the stream, or 0 */
	const uint32_t tail_n = (TAIL && SCAN == 2 && !REPAIR && !CUB && P.tail_ok) ? uni(fi.tail) : 0u;
	const uint32_t tail_s = tail_n ? uni(fi.tail_stream) : 0u;
	(void)tail_s;
	uint32_t cgm = cstart; /* the group number mod the look-back ring (groups cstart, cstart + waves, ...) */
	/* Round 6: the group's evaluation (k_fast_group.h) in two forms (SPLIT builds). A group that touches neither end of the segment
	 * -- all but two of a voice's hundreds or thousands -- has every frame of every row inside the segment, carries no state in and
	 * stages none out: in its form (EDGE false) the in-segment tests, the first- and last-group code and the masks behind them are
	 * compiled out. Those masks were most of the step interpreter's cost: per group a hundred scalar pairs computed, written to
	 * spill lanes (v_writelane) and read back in the steps (v_readlane) -- 3.3 of config 3's 36 vector instructions per
	 * operator-sample, 7 of the look-back builds' 47-69 (profiles/census/). The other form (EDGE true) is the code as it was, for the
	 * first and the last group, the repair pass and the builds without the split. A wave's groups ascend, so the edge groups are
	 * the first and the last of its run at most: three loops one after the other -- (first group) (groups between) (last group) --
	 * and not one loop with both forms in it, which kept the edge form's hoisted values alive across the other's iterations
	 * (178 spilled vector registers in the look-back build). */
#define FKG_ADVANCE() (it += gstride, cgm = cgm + wpv >= lk_ring ? cgm + wpv - lk_ring : cgm + wpv)
	if constexpr (SPLIT == 2 && !REPAIR) {
		/* (FK_INNER builds, fast_kernel: the groups between only -- the first and the last are another launch's, FastParams.edge_only) */
		for (uint32_t it = it_lo + cstart; it < n_iter; FKG_ADVANCE()) {
			if (it == 0 || it == last_group) continue;
			const uint32_t cg = it, repair_rows = 0;
			constexpr bool EDGE = false;
#include "k_fast_group.h"
		}
	} else if constexpr (SPLIT == 1 && !REPAIR) {
		uint32_t it = it_lo + cstart;
		if (it < n_iter && (it == 0 || it == last_group)) {
			const uint32_t cg = it, repair_rows = 0;
			constexpr bool EDGE = true;
#include "k_fast_group.h"
			FKG_ADVANCE();
		}
		for (; it < n_iter && it != last_group; FKG_ADVANCE()) {
			const uint32_t cg = it, repair_rows = 0;
			constexpr bool EDGE = false;
#include "k_fast_group.h"
		}
		if (it < n_iter) {
			const uint32_t cg = it, repair_rows = 0;
			constexpr bool EDGE = true;
#include "k_fast_group.h"
		}
	} else {
		for (uint32_t it = it_lo + cstart; it < n_iter; FKG_ADVANCE()) {
			const uint32_t cg = REPAIR ? uni(rep[2 + 2 * it]) : it;
			const uint32_t repair_rows = REPAIR ? uni(rep[3 + 2 * it]) : 0u;
			constexpr bool EDGE = true;
#include "k_fast_group.h"
		}
	}
#undef FKG_ADVANCE
	if (__any(zero_acc) && l == 0) atomicOr(&P.info[v].bail, 1u);

}

/* The mixer inside the closed-form launch (k_fast_types.h): tile j of chunk k, four frames per lane, every row of the stream
 * in voice order -- mix_kernel's sums (k_finish.h: mix_body, the tile whose rows all cover the segment with a constant pan;
 * generator.c:749-825) and its PCM. Two batches of INMIX_AHEAD 16-byte row loads in flight per lane: what a wave keeps in
 * flight is its bandwidth, and while it waits the SIMD's other three waves do not make up for it. */
#ifndef INMIX_AHEAD_N
#define INMIX_AHEAD_N 8 /* (a power of two, 64 at most) */
#endif
constexpr int INMIX_AHEAD = INMIX_AHEAD_N;
__device__ __forceinline__ uint32_t xcc_id() {
	uint32_t x;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
	return x & 7u;
}
/* (what the functions below need of FastParams) */
struct InmixArgs {
	uint32_t *ctl;
	const float *vout;
	const VoiceOut *vinfo;
	const MixStream *stream;
	uint32_t row_stride, flags, pcm_offset, nvc, div_m, div_s;
};
/* q / d for the divisor the host has prepared (m = floor(2^(32 + s) / d) + 1 - 2^32, s = ceil(log2 d); Granlund & Montgomery's
 * round-up form, exact for every 32-bit q): s_mul_hi_u32 and shifts -- a compiler-made 32-bit division is two dozen vector
 * instructions, and the task loop had five of them per task */
__device__ __forceinline__ uint32_t udiv_magic(const uint32_t q, const uint32_t m, const uint32_t s) {
	if (s == 0) return q;
	const uint32_t t = __umulhi(q, m);
	return (t + ((q - t) >> 1)) >> (s - 1);
}
typedef float __attribute__((ext_vector_type(4))) inmix_f4;
typedef uint32_t __attribute__((ext_vector_type(4))) inmix_u4;
__device__ __forceinline__ void inmix_tile(const InmixArgs &A, const uint32_t cs, const uint32_t ce /* the chunk's first frame, its end */,
		const uint32_t j, const int l) {
	const MixStream ms = *A.stream;
	const uint32_t i0 = cs + j * INMIX_TILE + 4u * (uint32_t)l; /* this lane's first frame */
	const uint32_t end = min(ce, ms.write_len);
	/* (a row's address: a buffer descriptor at the batch's first row + the row's offset in a scalar register + the lane's offset in
	 * ONE vector register for all the loads in flight. As plain pointers the compiler made per-lane 64-bit addresses of them, two
	 * registers per load. The host sees to it that INMIX_AHEAD rows span less than 4 GiB. Lanes past the chunk's end load its
	 * first frames and store nothing; a lane across the end loads up to three frames of the row beyond it -- row_stride is a
	 * multiple of 64 frames and no smaller than the stream -- and stores only its own.) */
	const uint32_t off = (i0 < end ? i0 : cs) * 4u;
	const float *rows = A.vout + (size_t)ms.first_row * A.row_stride;
	const VoiceOut *vo = A.vinfo + ms.first_row;
	/* (-DINMIX_DBG_SAME_ROW, -DINMIX_DBG_NOLOAD: timing aids -- every row the first one; no row loads at all. Wrong PCM.) */
#ifdef INMIX_DBG_SAME_ROW
	const uint32_t rbytes = 0u;
#else
	const uint32_t rbytes = A.row_stride * 4u;
#endif
	float L[4] = {0.f, 0.f, 0.f, 0.f}, R[4] = {0.f, 0.f, 0.f, 0.f};
	auto load = [&](inmix_f4 *sv, float &panv, uint32_t at) {
#ifdef INMIX_DBG_NOLOAD
#pragma unroll
		for (int u = 0; u < INMIX_AHEAD; ++u) { sv[u] = inmix_f4{(float)at, 1.f, 2.f, (float)l}; asm volatile("" : "+v"(sv[u])); }
		panv = 0.5f;
		return;
#endif
#ifdef INMIX_DBG_SAME_ROW
		const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)rows, 0, 0xffffffffu, 0x00020000);
#else
		const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(rows + (size_t)at * A.row_stride), 0, 0xffffffffu, 0x00020000);
#endif
#pragma unroll
		for (int u = 0; u < INMIX_AHEAD; ++u)
#ifndef FK_TEMPORAL_ROWS
			sv[u] = __builtin_bit_cast(inmix_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, (int)((uint32_t)u * rbytes), 2 /* nt */));
#else
			sv[u] = __builtin_bit_cast(inmix_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, (int)((uint32_t)u * rbytes), 0));
#endif
		/* (the rows' pans: one load -- lane u fetches row at + u's -- handed round by v_readlane; as loads of their own, one per
		 * row, the compiler made them vector loads and waited for each before the next) */
		panv = vo[at + ((uint32_t)l & (uint32_t)(INMIX_AHEAD - 1))].pan_const;
	};
	auto add = [&](const inmix_f4 *sv, const float panv) {
#pragma unroll
		for (int u = 0; u < INMIX_AHEAD; ++u) {
			const float pan = bits_f((uint32_t)__builtin_amdgcn_readlane((int)f_bits(panv), u));
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const float v = sv[u][q] * ms.amp_scale;
				const float s_r = v * pan;
				L[q] = (L[q] + v) - s_r;
				R[q] = (R[q] + v) + s_r;
			}
		}
	};
	inmix_f4 sa[INMIX_AHEAD], sb[INMIX_AHEAD];
	float pa = 0.f, pb = 0.f;
	uint32_t r = 0;
	if (r + INMIX_AHEAD <= ms.n_rows) load(sa, pa, r);
	while (r + INMIX_AHEAD <= ms.n_rows) {
		const bool more_b = r + 2 * INMIX_AHEAD <= ms.n_rows;
		if (more_b) load(sb, pb, r + INMIX_AHEAD);
		add(sa, pa);
		r += INMIX_AHEAD;
		if (!more_b) break;
		if (r + 2 * INMIX_AHEAD <= ms.n_rows) load(sa, pa, r + INMIX_AHEAD);
		add(sb, pb);
		r += INMIX_AHEAD;
	}
	for (; r < ms.n_rows; ++r) {
		const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(rows + (size_t)r * A.row_stride), 0, 0xffffffffu, 0x00020000);
		const inmix_f4 s1 = __builtin_bit_cast(inmix_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
		const float pan = vo[r].pan_const;
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const float v = s1[q] * ms.amp_scale;
			const float s_r = v * pan;
			L[q] = (L[q] + v) - s_r;
			R[q] = (R[q] + v) + s_r;
		}
	}
	const bool swap = (A.flags & 2u) != 0;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const uint32_t i = i0 + (uint32_t)q;
		if (i >= end) continue;
		if (A.flags & 1u) {
			int16_t *d = ms.pcm + 2 * (size_t)(A.pcm_offset + i);
			const int16_t l16 = pcm16(L[q]), r16 = pcm16(R[q]);
			d[0] = swap ? pcm_swap(l16) : l16;
			d[1] = swap ? pcm_swap(r16) : r16;
		} else {
			const int16_t m16 = pcm16((L[q] + R[q]) * 0.5f);
			ms.pcm[A.pcm_offset + i] = swap ? pcm_swap(m16) : m16;
		}
	}
}
/* (Inlined into fast_kernel's task loop, where the 12-row build fits its 128 vector registers exactly. Read from the kernel's
 * argument block behind a barrier the compiler cannot see through: taken from fast_kernel's copy of FastParams, these values
 * -- and what the tile derives from them, its row offsets -- were computed ahead of the task loop and kept across fast_voice:
 * 207 spilled scalars where 192 fit into three vector registers, and the fourth cost that build spilled vector registers in
 * its hot loops; as real calls the functions cost it 88 of them.) */
__device__ __forceinline__ InmixArgs inmix_args(const uint32_t nvc) {
	typedef const __attribute__((address_space(4))) FastParams *KArg;
	KArg Pk = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(Pk));
	InmixArgs A;
	A.ctl = Pk->inmix; A.vout = Pk->vout; A.vinfo = Pk->vinfo; A.stream = Pk->inmix_stream;
	A.row_stride = Pk->row_stride; A.flags = Pk->inmix_flags; A.pcm_offset = Pk->inmix_pcm_offset; A.nvc = nvc;
	A.div_m = Pk->inmix_div_m; A.div_s = Pk->inmix_div_s;
	return A;
}
__device__ __forceinline__ uint32_t inmix_ctl(const InmixArgs &A, uint32_t i) {
	return uni(__hip_atomic_load(&A.ctl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
/* the next task: from the XCD's own queue -- its chunks xcd, xcd + 8, ..., each for every voice -- then from the others'.
 * -> chunk << 32 | voice index, or ~0: none left */
__device__ __forceinline__ unsigned long long inmix_next(const InmixArgs &A, const uint32_t nch, int l) {
	const uint32_t xcd = xcc_id();
	for (uint32_t s_ = 0; s_ < 8; ++s_) {
		const uint32_t x = (xcd + s_) & 7u;
		const uint32_t mine = nch > x ? (nch - x + 7) / 8 : 0u; /* chunks of XCD x */
		/* (another XCD's queue: a look first, so that waves with nothing left do not keep adding to it) */
		if (s_ && inmix_ctl(A, INMIX_QUEUE + INMIX_LINE * x) >= mine * A.nvc) continue;
		uint32_t q = 0;
		if (l == 0) q = __hip_atomic_fetch_add(&A.ctl[INMIX_QUEUE + INMIX_LINE * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		q = uni(q);
		if (q < mine * A.nvc) {
			const uint32_t rank = udiv_magic(q, A.div_m, A.div_s);
			return ((unsigned long long)(x + 8 * rank) << 32) | (q - rank * A.nvc);
		}
	}
	return ~0ull;
}
/* after the task (voice index vi, chunk kc): its rows are counted; some of the chunk's tasks each mix tiles of the XCD's chunk before */
__device__ __forceinline__ void inmix_after(const InmixArgs &A, const uint32_t kc, const uint32_t vi, int l) {
	if ((kc & 7u) != xcc_id()) return; /* (a task from another XCD's queue: its rows are not in this L2, and that chunk stays incomplete) */
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wave's rows of the chunk are in the XCD's L2 */
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* (a compiler-visible order of the row stores before the count: ADVICE r05; no code another CU could see) */
	if (l == 0) __hip_atomic_fetch_add(&A.ctl[INMIX_CHUNK + INMIX_LINE * kc + INMIX_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	/* the chunk eight back, whose tiles this chunk's tasks mix: its frames (premix_kernel's words, from before the launch: plain
	 * loads) -- a chunk of the regular length, or one of the short ones at the end */
	const uint32_t cf = A.ctl[INMIX_CF], nch1 = A.ctl[INMIX_NCH1];
	uint32_t cs = 0, ce = 0;
	if (kc >= 8) {
		const uint32_t kb = kc - 8;
		if (kb < nch1) { cs = kb * cf; ce = min(cs + cf, A.ctl[INMIX_BASE]); }
		else { cs = A.ctl[INMIX_BASE] + (kb - nch1) * A.ctl[INMIX_CFS]; ce = cs + A.ctl[INMIX_CFS]; }
	}
	const uint32_t tpc = (ce - cs + INMIX_TILE - 1) / INMIX_TILE;
	/* Which tasks: those of voices part of the way into the chunk -- voice `first`: tile 0, the next: tile 1, ... (banks of fewer
	 * voices than tiles: every voice-count-th tile). Dealt out when the chunk before has been dealt out whole, they look at its
	 * counter a task's length later, when its last tasks have finished too; and they are through with their tiles before the
	 * chunk's last tasks end. (The chunk's LAST voices, first form: in the queue's last chunk those are the launch's last tasks,
	 * and their tiles went straight onto its end -- 0.1 ms of the 0.18 ms the mixing cost the launch.)
	 * (A second chance for a tile whose chunk was not whole then, from a task early in the chunk after, sixteen chunks on: tried;
	 * the launch got slower by more than the tiles were worth.) */
	const uint32_t first = A.nvc * ((A.flags >> 8) & 15u) / 16u; /* (sixteenths of the chunk: the host's choice) */
	uint32_t j = vi + A.nvc - first;
	if (j >= A.nvc) j -= A.nvc;
	if (kc < 8 || j >= tpc) return;
	if (inmix_ctl(A, INMIX_CHUNK + INMIX_LINE * (kc - 8) + INMIX_DONE) != A.nvc) return; /* (not all there after all: mix_kernel's) */
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* (this CU's L1 may hold older lines of the rows) */
	if (A.flags & 4u) return; /* (SAU_AMD_INMIX_DRY, a timing aid: everything but the tiles) */
	for (; j < tpc; j += A.nvc) {
		inmix_tile(A, cs, ce, j, l);
		if (l == 0) __hip_atomic_fetch_or(&A.ctl[INMIX_CHUNK + INMIX_LINE * (kc - 8) + INMIX_BITS + (j >> 5)], 1u << (j & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

#ifndef FK_SPLIT_MASK
#define FK_SPLIT_MASK 9 /* 1: closed form at 8 rows, 2: at 10, 4: at 12, 8: look-back at 8 (0 = none). Round 6, same box: config 4 4.37 -> 4.24 (8) -> 4.19 ms (9), FM bank 3.39 -> 3.13 ms (8); config 3 at 12 rows 1.99 -> 2.01 ms with 4 (175 spilled registers): not there (profiles/r06_ab.txt) */
#endif
/* SCAN: the kernel may meet voices with running-sum phases (it then holds both builds of fast_voice). */
#ifndef FK_MINB
#define FK_MINB 1
#endif
#ifdef FK_WAVES_EU /* tuning aid: a register budget that leaves room for another kernel's waves on the SIMD */
#define FK_ATTR __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(FK_WAVES_EU)))
#else
#define FK_ATTR __launch_bounds__(1024, FK_MINB)
#endif
/* TAIL: the look-back build that also mixes few-voice streams as it stores their last rows (FastParams.tail_ok) -- a build of its own:
 * with that code in it the look-back build spills 40-odd vector registers (0 without), 3 % of the FM bank's and config 4's launches */
/* INNER (round 6, the 12-row wide closed-form build -- BASELINE config 3's): only the row groups that touch neither end of the
 * segment, in the form without in-segment masks (fast_voice: EDGE false), one loop; the first and the last group of every voice are
 * rendered by a launch of the plain build ahead of this one (FastParams.edge_only). The three-loop form of fast_voice, which holds
 * both forms in one kernel, spilled 175 vector registers at 12 rows. */
template <int T, int SCAN, bool CUB = false, bool WIDE = false, bool TAIL = false, bool INNER = false>
__global__ void FK_ATTR fast_kernel(FastParams P) {
	static_assert(!INNER || (SCAN == 0 && WIDE && T == 12 && !CUB && !TAIL), "the inner-groups-only form exists for the 12-row wide closed-form build");
	static_assert(!WIDE || ((SCAN == 0 || SCAN == 2) && !CUB), "only the closed-form and the look-back builds have a wide-table form");
	static_assert(!TAIL || (SCAN == 2 && !CUB && T == 8), "the stream-mixing form exists for the 8-row look-back build");
	constexpr int NP = 64 * T;
	constexpr int W = 16;
	/* the builds whose groups away from the segment's ends take a copy of their own (fast_voice: SPLIT) -- the ones BASELINE's
	 * configurations and the FM bank run in; each costs its compile time and code size twice */
	constexpr int SPLIT = INNER ? 2 : (!CUB && ((SCAN == 0 && T == 8 && (FK_SPLIT_MASK & 1)) || (SCAN == 0 && T == 10 && (FK_SPLIT_MASK & 2)) ||
	                                (SCAN == 0 && T == 12 && (FK_SPLIT_MASK & 4)) || (SCAN == 2 && T == 8 && (FK_SPLIT_MASK & 8)))) ? 1 : 0;
	extern __shared__ __align__(16) unsigned char lds[];
	const int tid = threadIdx.x;
	const int w = (int)uni((uint32_t)tid >> 6);
	const int l = tid & 63;
	/* a sum pass nobody needs costs a launch, not a table staging */
	if (SCAN == 1 && P.mode != 0 && P.mode <= P.sum_levels && P.pass_flags[P.mode - 1] == 0) return;
	if (SCAN == 1 && P.mode == P.sum_levels + 2 && P.pass_flags[FAST_MAX_LEVELS + 1] == 0) return; /* no chains */
	if (SCAN == 1 && P.only_multi && P.pass_flags[FAST_MAX_LEVELS + 2] == 0) return; /* no voice the single-pass build left out */
	if (SCAN == 3 && P.pass_flags[FAST_LEAN_FLAG] == 0) return; /* no voice with chains and nothing to scan */
	if (SCAN == 0 && P.split_cf && P.pass_flags[FAST_CF_COUNT] == 0) return; /* no closed-form voice beside the look-back ones */
	if (CUB && P.pass_flags[FAST_CUB_FLAG] == 0) return; /* no voice for the build with the `cub` tails */
	if (SCAN == 2 && P.pass_flags[FAST_LK_COUNT] == 0) return; /* ... and the other way round */

	const uint32_t tabs = (uint32_t)(uintptr_t)lds;
	unsigned char *areas = lds + (size_t)P.n_tabs * FkTab<WIDE>::BYTES;
	const size_t area_bytes = (size_t)P.n_fast * NP * sizeof(float) + (size_t)P.max_steps * sizeof(unsigned long long);
	float *slots = (float *)(areas + (size_t)w * area_bytes) + l; /* lane's column of every row */
	unsigned long long *carry = (unsigned long long *)(areas + (size_t)w * area_bytes + (size_t)P.n_fast * NP * sizeof(float)); /* per step */
	unsigned long long *lring = nullptr; /* the single-pass build: look-back rings after the waves' areas, zeroed */
	if (SCAN == 2) {
		lring = (unsigned long long *)(areas + (size_t)W * area_bytes);
		lring[tid] = 0;
		static_assert(LOOK_LDS_BYTES == 1024 * sizeof(unsigned long long), "one word per thread");
	}

	fk_stage_tables<WIDE>(P, lds, (uint32_t)tid, 64 * W);
	__syncthreads(); /* the only barrier: tables are shared, all else is per wave */

	const uint32_t NV = P.n_voices;
	if (SCAN == 0) {
		/* The closed-form build: task = (voice, one of dyn_chunks runs of consecutive row groups), in voice order.
		 * Dealt out by a counter, or -- dyn_static -- in fixed strides over the launch's waves. One loop, one copy of
		 * fast_voice, and the atomic's answer is waited for where it is asked for (the SIMD's other waves run meanwhile):
		 * a second call site, or the answer held in a vector register across a task so as to ask a task ahead, cost
		 * the 8-row build 60 spilled VGPRs, 2.6 GB of scratch traffic per launch and 16 % more VALU instructions (r03). */
		const uint32_t NVc = P.split_cf ? P.pass_flags[FAST_CF_COUNT] : NV; /* (split: the voices of vlists[0]) */
		const uint32_t K = P.dyn_chunks ? P.dyn_chunks : 1u, n_tasks = NVc * K;
		const bool counted = P.dyn_static == 0;
		const uint32_t stride = gridDim.x * W;
		uint32_t snext = blockIdx.x * W + (uint32_t)w;
		/* tasks from the XCDs' queues (k_fast_types.h), and the launch mixes when the host asks for it and premix_kernel has found
		 * nothing against it */
		const bool im = !CUB && (P.inmix_flags & 64u) != 0;
		const bool mixing = im && (P.inmix_flags & 32u) && uni(__hip_atomic_load(&P.work_count[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0;
		for (;;) {
			uint32_t t_ = snext;
			if (counted && !im) { t_ = 0; if (l == 0) t_ = atomicAdd(&P.pass_flags[FAST_DYN_CTR], 1u); }
			if (!CUB && im) { /* (as a task's number in the one queue's order: voice x chunks + chunk) */
				const InmixArgs A = inmix_args(NVc);
				const unsigned long long nx = inmix_next(A, mixing ? A.ctl[INMIX_NCH] : K, l);
				t_ = nx == ~0ull ? n_tasks : (uint32_t)nx * K + (uint32_t)(nx >> 32);
			}
			const uint32_t task = uni(t_);
			if (task >= n_tasks) break;
			snext = task + stride;
			const uint32_t vi = task / K;
			const uint32_t v = P.split_cf ? P.vlists[vi] : vi;
			const FastInfo fi = P.info[v];
			fast_voice<T, 0, false, CUB, WIDE, SPLIT>(P, v, fi, slots, carry, tabs, l, 1u, 0u, nullptr, task - vi * K, K);
			if (!CUB && mixing) {
				const InmixArgs A = inmix_args(NVc);
				inmix_after(A, task - vi * K, vi, l);
			}
		}
		return;
	}
	const uint32_t g = blockIdx.x * W + (uint32_t)w;
	const uint32_t total_waves = gridDim.x * W;
	if (SCAN == 2) {
		/* (this build's launches always come with the lists) the look-back voices of vlists[1]: waves per voice from how many there are -- as many as a voice has row
		 * groups, 64 at most, where the words in HBM exist; 16, 8, 4, 2 or 1 keep a voice inside a workgroup (rings in LDS) */
		const uint32_t NVl = P.pass_flags[FAST_LK_COUNT];
		uint32_t wpv = P.look_wpv;
		if (P.look_words_real) {
			wpv = total_waves / NVl;
			if (wpv > P.look_groups) wpv = P.look_groups;
			wpv = wpv >= 64 ? 64u : wpv >= 32 ? 32u : wpv >= 16 ? 16u : wpv >= 8 ? 8u : wpv >= 4 ? 4u : wpv >= 2 ? 2u : 1u;
		}
		const uint32_t slots_v = total_waves / wpv; /* voices in flight at once */
		/* Which wave takes which voice: neighbouring waves, so that a voice inside a workgroup (wpv <= 16, rings in LDS) has its waves
		 * on different SIMDs, and a voice spread over workgroups (wpv 32 or 64, words in HBM; BASELINE config 4: 64 voices on 4096
		 * waves) sits in 2 or 4 neighbouring ones. Round 6 tried the other placement for those -- wave g takes voice slot g mod slots_v,
		 * group phase g / slots_v, a workgroup's 16 waves 16 different voices', so that a wave waiting for its voice's sums leaves the
		 * SIMD to three that do not: config 4 4.39 -> 4.60 ms per step. A round's sums are then gathered from 64 CUs instead of 4,
		 * and every look-back waits for the slowest of them (look_wpv_flags & 4: SAU_AMD_LOOK_SPREAD, kept for the record;
		 * profiles/r06_ab.txt). */
		const bool spread = wpv >= 32 && (P.look_wpv_flags & 4u);
		const uint32_t j0 = spread ? g % slots_v : g / wpv, cs = spread ? g / slots_v : g % wpv;
		if (j0 >= slots_v || cs >= wpv) return; /* (waves beyond the last whole voice) */
		for (uint32_t j = j0; j < NVl; j += slots_v) {
			const uint32_t v = P.vlists[NV + j];
			const FastInfo fi = P.info[v];
			fast_voice<T, 2, false, CUB, WIDE, SPLIT, TAIL>(P, v, fi, slots, carry, tabs, l, wpv, cs, lring, 0u, 0u, j);
		}
		return;
	}
	uint32_t wpv = total_waves >= NV ? total_waves / NV : 1; /* waves per voice */
	uint32_t v = total_waves >= NV ? g / wpv : g;
	const uint32_t vstride = total_waves >= NV ? NV : total_waves;
	const uint32_t cstart = total_waves >= NV ? g % wpv : 0;

	for (; v < NV; v += vstride) {
		const FastInfo fi = P.info[v];
		const uint32_t seq_kind = SCAN ? uni(fi.seq) : 0u;
		const bool lean_voice = SCAN && P.lean_on && seq_kind == 2 && uni(fi.n_scan) == 0; /* the build of its own takes it */
		if (SCAN == 3) {
			if (P.mode == P.sum_levels + 2 && uni(fi.n_chain) == 0) continue; /* (its chains are fed by chain_kernel itself) */
			if (lean_voice) fast_voice<T, 3>(P, v, fi, slots, carry, tabs, l, wpv, cstart);
			continue;
		}
		if (SCAN == 1 && lean_voice) continue;
		if (SCAN == 1 && P.mode != 0 && P.mode <= P.sum_levels && (seq_kind != 2 || uni(fi.levels) < P.mode))
			continue; /* a sum pass only concerns multi-pass voices that deep */
		if (SCAN == 1 && P.mode == P.sum_levels + 2 && (seq_kind != 2 || uni(fi.n_chain) == 0))
			continue; /* the chain-input pass only concerns voices with feedback chains */
		if (SCAN == 1 && (P.only_multi ? (seq_kind != 1 && seq_kind != 2) : seq_kind == 3)) continue;
		if (SCAN && seq_kind != 0) fast_voice<T, 1>(P, v, fi, slots, carry, tabs, l, wpv, cstart);
		else fast_voice<T, 0>(P, v, fi, slots, carry, tabs, l, wpv, cstart);
	}
}

/* Round 6: the closed-form voices and the look-back voices of a segment in ONE launch. Apart, the
 * look-back launch leaves a fifth of its issue slots empty: twice per row group a wave waits for the sums of the groups before
 * it (words in HBM: a round trip of microseconds), and a workgroup's sixteen waves are one voice's and wait together -- nothing
 * on the SIMD fills the gap (SQ_ACTIVE_INST_VALU 0.80; BASELINE config 4: 2.05 ms for the look-back voices + 1.35 ms for the
 * closed-form ones). Here some of a workgroup's waves -- half of them for config 4 -- render look-back voices (as fast_kernel<8, 2>:
 * waves per voice from how many such voices there are) at a raised priority, and the others take closed-form tasks from the queues
 * (as fast_kernel<8, 0>): every SIMD holds waves of both kinds, and what the waiting ones leave is the closed-form waves'. One copy of the tables;
 * every wave's block buffers are sized for the look-back voices', so that when analyze_kernel finds no voice of one kind all
 * sixteen waves render the other. P: the look-back launch's parameters, Q: the closed-form launch's (both at 8 rows per pass,
 * narrow tables). */
__global__ void __launch_bounds__(1024, 1) duo_kernel(FastParams P, FastParams Q) {
	constexpr int T = 8, NP = 64 * T, W = 16;
	constexpr int SPLIT8 = (FK_SPLIT_MASK & 8) ? 1 : 0, SPLIT1 = (FK_SPLIT_MASK & 1) ? 1 : 0;
	extern __shared__ __align__(16) unsigned char lds[];
	const int tid = threadIdx.x;
	const int w = (int)uni((uint32_t)tid >> 6);
	const int l = tid & 63;
	const uint32_t NVl = P.pass_flags[FAST_LK_COUNT], NVc = Q.pass_flags[FAST_CF_COUNT];
	if (NVl == 0 && NVc == 0) return;
	const uint32_t tabs = (uint32_t)(uintptr_t)lds;
	unsigned char *areas = lds + (size_t)P.n_tabs * FkTab<false>::BYTES;
	const size_t area_bytes = (size_t)P.n_fast * NP * sizeof(float) + (size_t)P.max_steps * sizeof(unsigned long long); /* (P.n_fast >= Q.n_fast) */
	float *slots = (float *)(areas + (size_t)w * area_bytes) + l;
	unsigned long long *lring = (unsigned long long *)(areas + (size_t)W * area_bytes);
	lring[tid] = 0;
	fk_stage_tables<false>(P, lds, (uint32_t)tid, 64 * W);
	__syncthreads();
	const uint32_t NV = P.n_voices;
	/* which waves do what: by the lists' lengths (all of one kind when the other has no voice). Waves 0 .. lw - 1: look-back voices.
	 * (BASELINE config 4, as many voices of one kind as of the other, ms per step by lw: 5: 5.43, 6: 4.71, 7: 4.16, 8: 3.77, 9: 3.77,
	 * 10: 4.01, 12: 5.07, 14: 8.30 -- apart: 4.03; profiles/r06_ab.txt. Too few look-back waves and each has too many row groups to
	 * walk one after the other; too few closed-form waves and they cannot issue what the launch leaves them) */
	uint32_t lw = NVc == 0 ? (uint32_t)W : 0u;
	if (NVl && NVc) {
		lw = (uint32_t)((16ull * NVl + (NVl + NVc) / 2) / ((unsigned long long)NVl + NVc));
		lw = lw < 2 ? 2u : lw > 14 ? 14u : lw;
		if ((P.look_wpv_flags >> 8) & 15u) lw = (P.look_wpv_flags >> 8) & 15u; /* (SAU_AMD_DUO_LW: a tuning aid) */
	}
	if ((uint32_t)w < lw) {
		if (lw < (uint32_t)W) { /* (the waves whose waits set the launch's length go first; look_wpv_flags bits 12-14: SAU_AMD_DUO_PRIO, a tuning aid) */
			const uint32_t pr = (P.look_wpv_flags >> 12) & 7u;
			if (pr == 0 || pr == 3) __builtin_amdgcn_s_setprio(2);
			else if (pr == 2) __builtin_amdgcn_s_setprio(1);
			else if (pr == 4) __builtin_amdgcn_s_setprio(3);
		}
		unsigned long long *carry = (unsigned long long *)(areas + (size_t)w * area_bytes + (size_t)P.n_fast * NP * sizeof(float));
		const uint32_t g = blockIdx.x * lw + (uint32_t)w, total_waves = gridDim.x * lw;
		uint32_t wpv = total_waves / NVl; /* (this launch always comes with the words in HBM: a voice may spread over workgroups) */
		if (wpv > P.look_groups) wpv = P.look_groups;
		/* (a voice spread over workgroups looks back through the words in HBM and may have any number of waves up to 64 -- the
		 * lanes of one poll --; one inside a workgroup a power of two: its rings in LDS) */
		wpv = wpv >= 64 ? 64u : wpv > 16 ? wpv : wpv >= 16 ? 16u : wpv >= 8 ? 8u : wpv >= 4 ? 4u : wpv >= 2 ? 2u : 1u;
		const uint32_t slots_v = total_waves / wpv;
		const uint32_t j0 = g / wpv, cs = g % wpv;
		if (j0 >= slots_v) return;
		for (uint32_t j = j0; j < NVl; j += slots_v) {
			const uint32_t v = P.vlists[NV + j];
			const FastInfo fi = P.info[v];
			fast_voice<T, 2, false, false, false, SPLIT8, false>(P, v, fi, slots, carry, tabs, l, wpv, cs, lring, 0u, 0u, j, lw);
		}
		return;
	}
	/* closed-form tasks: fast_kernel<8, 0>'s loop over Q (tasks from the XCDs' queues or the one counter, or in fixed strides) */
	unsigned long long *carry = (unsigned long long *)(areas + (size_t)w * area_bytes + (size_t)Q.n_fast * NP * sizeof(float));
	const uint32_t cw = (uint32_t)W - lw; /* closed-form waves per workgroup */
	const uint32_t K = Q.dyn_chunks ? Q.dyn_chunks : 1u, n_tasks = NVc * K;
	const bool counted = Q.dyn_static == 0;
	const uint32_t stride = gridDim.x * cw;
	uint32_t snext = blockIdx.x * cw + ((uint32_t)w - lw);
	const bool im = (Q.inmix_flags & 64u) != 0;
	for (;;) {
		uint32_t t_ = snext;
		if (counted && !im) { t_ = 0; if (l == 0) t_ = atomicAdd(&Q.pass_flags[FAST_DYN_CTR], 1u); }
		if (im) {
			InmixArgs A;
			A.ctl = Q.inmix; A.vout = Q.vout; A.vinfo = Q.vinfo; A.stream = Q.inmix_stream; A.row_stride = Q.row_stride;
			A.flags = Q.inmix_flags; A.pcm_offset = Q.inmix_pcm_offset; A.nvc = NVc; A.div_m = Q.inmix_div_m; A.div_s = Q.inmix_div_s;
			const unsigned long long nx = inmix_next(A, K, l);
			t_ = nx == ~0ull ? n_tasks : (uint32_t)nx * K + (uint32_t)(nx >> 32);
		}
		const uint32_t task = uni(t_);
		if (task >= n_tasks) break;
		snext = task + stride;
		const uint32_t vi = task / K;
		const uint32_t v = Q.vlists[vi];
		const FastInfo fi = Q.info[v];
		fast_voice<T, 0, false, false, false, SPLIT1>(Q, v, fi, slots, carry, tabs, l, 1u, 0u, nullptr, task - vi * K, K);
	}
}

/* The row groups fast_kernel noted (see FAST_REPAIR_SHIFT): same workgroup shape and LDS layout. */
template <int T, bool WIDE = false>
__global__ void __launch_bounds__(1024) repair_kernel(FastParams P) {
	constexpr int NP = 64 * T;
	constexpr int W = 16;
	extern __shared__ __align__(16) unsigned char lds[];
	if (P.pass_flags[FAST_MAX_LEVELS] == 0) return; /* the usual case */
	const int tid = threadIdx.x;
	const int w = (int)uni((uint32_t)tid >> 6);
	const int l = tid & 63;
	const uint32_t tabs = (uint32_t)(uintptr_t)lds;
	unsigned char *areas = lds + (size_t)P.n_tabs * FkTab<WIDE>::BYTES;
	const size_t area_bytes = (size_t)P.n_fast * NP * sizeof(float) + (size_t)P.max_steps * sizeof(unsigned long long);
	float *slots = (float *)(areas + (size_t)w * area_bytes) + l;
	unsigned long long *carry = (unsigned long long *)(areas + (size_t)w * area_bytes + (size_t)P.n_fast * NP * sizeof(float));
	fk_stage_tables<WIDE>(P, lds, (uint32_t)tid, 64 * W);
	__syncthreads();
	/* one wave per voice with noted groups */
	for (uint32_t v = blockIdx.x * W + (uint32_t)w; v < P.n_voices; v += gridDim.x * W) {
		if (uni(P.repair[(size_t)v * FAST_REPAIR_WORDS]) == 0) continue;
		const FastInfo fi = P.info[v];
		if (uni(fi.total) == 0 || uni(fi.seq) != 0) continue;
		fast_voice<T, 0, true, false, WIDE>(P, v, fi, slots, carry, tabs, l, 1u, 0u);
	}
}
