/* k_chain.h -- part of hip_backend.hip (included there, inside namespace sauhip; not a header of its own).
 * chain_kernel: feedback recurrences with lanes = voices. */
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
#ifndef SAU_CHAIN_BATCH_FRAMES
#define SAU_CHAIN_BATCH_FRAMES 32
#endif
/* frames per lane and batch: what a batch costs beside the recurrence itself (its LDS reads and writes, the bound check
 * of the short rounding form, the barrier) is spread over that many sample steps -- 16: 117 ns per step, 32: see DESIGN.md 4.3 */
constexpr uint32_t CHAIN_BATCH = SAU_CHAIN_BATCH_FRAMES;
constexpr uint32_t CHAIN_NQ = CHAIN_BATCH / 4;     /* 16-byte quads of a batch */
constexpr uint32_t CHAIN_IO_WORDS = CHAIN_BATCH * 64; /* one array of one batch */
constexpr uint32_t CHAIN_TAB_BYTES = 65536, CHAIN_TAB_C01 = 32768; /* a wave table in chain_kernel's LDS: [c3, c2] x 2048, then [c1, c0] x 2048 as f64 */
constexpr size_t CHAIN_IO_BYTES = (size_t)(2 * 2 + 2) * CHAIN_IO_WORDS * 4; /* in[2][2] + out[2] */

/* LDS layout of a batch array: frame 4q + r of lane l at word (q * 64 + l) * 4 + r -- a lane's four 16-byte
 * accesses are conflict-free */
__device__ __forceinline__ uint32_t chain_io_word(uint32_t q, int l) { return (q * 64u + (uint32_t)l) * 4u; }

/* SMALL: every feedback offset of the batch is known to stay below 2^20 cycles in magnitude, where the short
 * rounding form is exact (rint32w_p31_small) -- no per-sample test on the chain; the caller verifies the bound
 * it assumed for |fb_s| afterwards (fb_max) and redoes the batch without SMALL if it was exceeded. */
template <bool LDS_TAB, bool TAIL, bool SMALL, int INL /* 0: no lane sums increments, 1: every lane does, 2: some do (`inl`) */>
__device__ __forceinline__ void chain_batch(const uint4 *bq, const float4 *aq, float4 *sq, uint32_t t, uint32_t n,
		uint32_t tab23, uint32_t tab01, const HerpC23 *g23, const HerpC01 *g01, float dscale, float doff,
		uint32_t &prev_phase, double &prev_Is, float &prev_s, float &fb_s, float &fb_max, const bool inl, uint32_t &acc) {
	/* two 16-byte LDS reads per sample from one address: the table block holds [c3, c2] and, 32 KiB further on, [c1, c0]
	 * widened to f64 when it was staged (CHAIN_TAB_BYTES) -- no conversion and one address computation less on and
	 * beside the dependent chain: 110.5 -> 96.9 ns per step in the bare loop (tools/chain_probe2.hip, round 4) */
	typedef double __attribute__((ext_vector_type(2))) f64x2;
	typedef const f64x2 __attribute__((address_space(3))) *lds_f64x2;
#pragma unroll
	for (int u = 0; u < (int)CHAIN_NQ; ++u) {
		const uint32_t b4[4] = {bq[u].x, bq[u].y, bq[u].z, bq[u].w};
		const float a4[4] = {aq[u].x, aq[u].y, aq[u].z, aq[u].w};
		float s4[4], fbn[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const float p = fb_s * a4[j];
			uint32_t ofs = rint32w_p31_small(p);
			if (!SMALL) { if (__builtin_expect(!(fabsf(p) < 0x1p20f), 0)) ofs = rint32w(p * 0x1p31f); }
			/* (inline chains: the rows of LDS hold increments, summed here -- beside the dependent chain, not on it) */
			const uint32_t acc_n = INL ? acc + b4[j] : 0u;
			const uint32_t phase = (INL == 1 ? acc_n : INL == 2 ? (inl ? acc_n : b4[j]) : b4[j]) + ofs;
			const int32_t d = (int32_t)(phase - prev_phase);
			uint32_t ind = phase >> SLEN_BITS;
			double Isv;
			if (LDS_TAB) {
				/* (two instructions on the chain, v_lshrrev + v_lshl_add: the empty asm keeps LLVM from rewriting the address as
				 * ((phase >> 17) & 0x7ff0) + tab, which is three -- as in fk_entry, k_fast_voice.h) */
				asm("" : "+v"(ind));
				const uint32_t a = tab23 + (ind << 4);
				const f64x2 c32 = *(lds_f64x2)(uintptr_t)a;
				const f64x2 c10 = *(lds_f64x2)(uintptr_t)(a + CHAIN_TAB_C01);
				const double x = (double)(phase & (SLEN - 1)); /* herp_poly (sau_dev_math.h), the same operations in the same order */
				Isv = ((c32.x * x + c32.y) * x + c10.x) * x + c10.y;
			} else {
				Isv = herp_poly(g23[ind], g01[ind], phase);
			}
			const float sv_new = wosc_diff(Isv, prev_Is, d, dscale, doff);
			bool hold = d == 0; /* wosc.h:292-293: a repeated phase holds the previous sample */
			const bool act = !TAIL || t + (uint32_t)(4 * u + j) < n;
			if (INL) acc = (TAIL && !act) ? acc : acc_n; /* (wosc.h:145: pre-increment; stands still past the chain's last frame) */
			if (TAIL) hold = hold || !act;
			const float sv = hold ? prev_s : sv_new;
			/* (a repeated phase has the same Hermite value -- prev_Is is always the value at prev_phase: wosc.h:215-231 and
			 * 247-262 keep them together -- so only a lane past its chain's end needs the select: two instructions per step) */
			prev_Is = TAIL ? (hold ? prev_Is : Isv) : Isv;
			prev_phase = (TAIL && !act) ? prev_phase : phase; /* (equal to the old one when held) */
			prev_s = sv;
			s4[j] = sv;
			const float fb_n = (fb_s + sv) * 0.5f;
			fb_s = (TAIL && !act) ? fb_s : fb_n;
			fbn[j] = fb_n;
		}
		/* the bound the short rounding form assumed, beside the chain: one v_max3_f32 per two steps. It passes a NaN by, but a
		 * NaN or an infinity in fb_s stays there (the average of it and anything is it again, or a NaN), so the caller's look
		 * at the batch's last fb_s finds those */
		if (SMALL) {
			asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(fb_max) : "v"(fbn[0]), "v"(fbn[1]));
			asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(fb_max) : "v"(fbn[2]), "v"(fbn[3]));
		}
		sq[u] = make_float4(s4[0], s4[1], s4[2], s4[3]);
	}
}

/* Sixteen consecutive values of a line, frames [t, t + 16) of the segment. Lanes hold different lines: the shape
 * is tested once per batch and shape (not once per value), each shape's loop compiled with its type known. */
template <uint32_t TYPE, uint32_t N>
__device__ __forceinline__ void line_batch_shape(const FastLine &fl, uint32_t t, float *out) {
	if (fl.sw.type != TYPE) return;
	Sweep sw = fl.sw;
	sw.type = TYPE;
	/* (every value computed and then selected -- one straight run of independent polynomials the scheduler can interleave;
	 * with a branch per value the feeder wave, alone on its SIMD, paid every instruction's latency: round 4) */
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) {
		const float v = sweep_value_inl<true>(sw, t + j);
		out[j] = t + j < fl.goal_len ? v : out[j];
	}
}
/* xpe / lge (what `exp` and `log` sweeps resolve to, sau/line.c:125-148: the usual glide) stage by stage over the batch:
 * sweep_value_inl's operations in its order, value by value the same -- but written as thirteen runs of independent
 * instructions, so that the wave, alone on its SIMD, issues back to back instead of waiting out each value's dependent
 * chain (round 4: the value-by-value form cost the inline feeder 4 us of a 32-frame batch, the chain wave needs 3.4) */
template <bool XPE, uint32_t N>
__device__ __forceinline__ void line_batch_exp(const FastLine &fl, uint32_t t, float *out) {
	if (fl.sw.type != (XPE ? LN_xpe : LN_lge)) return;
	const Sweep sw = fl.sw;
	const float d = XPE ? (sw.v0 - sw.vt) : (sw.vt - sw.v0), base = XPE ? sw.vt : sw.v0;
	float x[N], x2[N], x3[N], p[N];
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) x[j] = (float)(t + j + sw.pos) * sw.inv_time;
	if (XPE) {
#pragma unroll
		for (uint32_t j = 0; j < N; ++j) x[j] = 1.f - x[j];
	}
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) x2[j] = x[j] * x[j];
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) x3[j] = x2[j] * x[j];
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) p[j] = x[j] * (629.f / 1792.f);
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) x[j] = x2[j] * (1163.f / 1792.f);
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) p[j] = p[j] + x[j];
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) x[j] = x3[j] + -1.f;
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) p[j] = p[j] * x[j];
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) p[j] = p[j] * x2[j];
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) p[j] = x3[j] + p[j];
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) p[j] = d * p[j];
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) p[j] = base + p[j];
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) out[j] = t + j < fl.goal_len ? p[j] : out[j];
}
template <uint32_t N>
__device__ __forceinline__ void line_batch(const FastLine &fl, uint32_t t, float *out) {
#pragma unroll
	for (uint32_t j = 0; j < N; ++j) out[j] = fl.hold;
	if (t >= fl.goal_len) return;
	line_batch_shape<LN_cos, N>(fl, t, out); line_batch_shape<LN_lin, N>(fl, t, out); line_batch_shape<LN_sah, N>(fl, t, out);
	line_batch_exp<true, N>(fl, t, out); line_batch_exp<false, N>(fl, t, out); line_batch_shape<LN_sqe, N>(fl, t, out);
	line_batch_shape<LN_cub, N>(fl, t, out); line_batch_shape<LN_smo, N>(fl, t, out); line_batch_shape<LN_ncl, N>(fl, t, out);
	line_batch_shape<LN_nhl, N>(fl, t, out); line_batch_shape<LN_uwh, N>(fl, t, out);
	/* (LN_exp / LN_log were resolved to xpe / lge when the sweep was set up: sau/line.c:125-148) */
}

/* the feeder's share of one batch: inputs of frames [t, t + 16) into the LDS arrays */
__device__ __forceinline__ void chain_feed(const ChainDesc &cd, bool live, int l, uint32_t t, uint32_t *acc,
		const uint4 *bp, const float4 *ap, uint32_t *in_base, float *in_amt, const uint32_t half) {
	if (!live) return;
	uint32_t a = *acc, a_end = *acc; /* a_end: the accumulator after the segment's last frame, should it fall in this batch */
	if (chain_mode(cd) == CM_INLINE) {
		/* Frequency and amounts from the operator's own lines (sau/line.c fills are functions of the position). Two feeder
		 * waves share a batch, half its frames each (`half`): a wave alone on its SIMD pays its instructions' latencies, and
		 * one wave took 5 us for the 32 frames the chain wave consumes in 3.4 (round 4). What goes to LDS are phase
		 * *increments*: the chain wave sums them itself (chain_batch: `inl`), so neither half waits for the other's sum. */
		constexpr uint32_t HB = CHAIN_BATCH / 2, HQ = CHAIN_NQ / 2;
		{
			/* The usual case in one straight run, no shape dispatch and no test per value: every lane's amount line is a `lin`
			 * ramp that covers this half batch, or holds; its frequency is constant, or an xpe / lge glide (what `exp` and `log`
			 * resolve to) that covers it, or holds; no ratio multiplier. The same operations per value as line_batch's, in its
			 * order. (The feeders' busy time is what slows the chain wave beside them, in proportion: about a thousand
			 * instructions per feeder and batch through the general code below, 450 here.) */
			const uint32_t tb = t + half * HB;
			const bool p_hold = tb >= cd.pl.goal_len, p_in = cd.pl.sw.type == LN_lin && tb + HB <= cd.pl.goal_len;
			const bool fconst = (cd.lflags & CL_FCONST) != 0;
			const bool f_hold = tb >= cd.fl.goal_len;
			const bool xpe = cd.fl.sw.type == LN_xpe;
			const bool f_in = (xpe || cd.fl.sw.type == LN_lge) && tb + HB <= cd.fl.goal_len;
			const bool any_mul = (cd.lflags & (CL_MUL_GOAL | CL_MUL_HOLD)) != 0;
			if (__all((p_hold || p_in) && (fconst || ((f_hold || f_in) && !any_mul)))) {
				float m[HB];
				uint32_t incs[HB];
				{
					const Sweep sw = cd.pl.sw;
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) {
						const float v = sw.vm + sw.k * (float)((int32_t)(tb + j) + sw.adj_pos); /* (sweep_value_inl: LN_lin) */
						m[j] = p_hold ? cd.pl.hold : v;
					}
				}
				if (!__all(fconst)) {
					const Sweep sw = cd.fl.sw;
					const float d = xpe ? (sw.v0 - sw.vt) : (sw.vt - sw.v0), base = xpe ? sw.vt : sw.v0;
					float x[HB], x2[HB], x3[HB], pp[HB];
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) x[j] = (float)(tb + j + sw.pos) * sw.inv_time;
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) x[j] = xpe ? 1.f - x[j] : x[j];
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) x2[j] = x[j] * x[j];
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) x3[j] = x2[j] * x[j];
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) pp[j] = x[j] * (629.f / 1792.f);
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) x[j] = x2[j] * (1163.f / 1792.f);
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) pp[j] = pp[j] + x[j];
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) x[j] = x3[j] + -1.f;
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) pp[j] = pp[j] * x[j];
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) pp[j] = pp[j] * x2[j];
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) pp[j] = x3[j] + pp[j];
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) pp[j] = d * pp[j];
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) pp[j] = base + pp[j];
					bool big = false;
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) {
						x[j] = cd.coeff * (f_hold ? cd.fl.hold : pp[j]);
						big |= !(fabsf(x[j]) < 0x1p50f);
					}
					if (!__any(big)) {
#pragma unroll
						for (uint32_t j = 0; j < HB; ++j) incs[j] = (uint32_t)__double2loint((double)x[j] + 0x1.8p52);
					} else {
#pragma unroll
						for (uint32_t j = 0; j < HB; ++j) incs[j] = rint32w(x[j]);
					}
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) incs[j] = fconst ? cd.inc_const : incs[j];
				} else {
#pragma unroll
					for (uint32_t j = 0; j < HB; ++j) incs[j] = cd.inc_const;
				}
#pragma unroll
				for (uint32_t q = 0; q < HQ; ++q) {
					*(uint4 *)(in_base + chain_io_word(half * HQ + q, l)) = make_uint4(incs[4 * q], incs[4 * q + 1], incs[4 * q + 2], incs[4 * q + 3]);
					*(float4 *)(in_amt + chain_io_word(half * HQ + q, l)) = make_float4(m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]);
				}
				return;
			}
		}
		/* (four frames at a time in a real loop: the instruction cache is shared with the chain wave, whose 32 unrolled steps
		 * are 8 KiB by themselves, and with the CU next door -- the feeders' lines unrolled over sixteen frames made the
		 * chain wave 5 % slower without it ever waiting at the barrier: -DCHAIN_PROF, round 4) */
#pragma unroll 1
		for (uint32_t q = 0; q < HQ; ++q) {
			const uint32_t th = t + half * HB + 4 * q;
			float fv[4], m[4];
			uint32_t incs[4];
			line_batch<4>(cd.pl, th, m);
			if (!(cd.lflags & CL_FCONST)) {
				line_batch<4>(cd.fl, th, fv);
				float x[4];
				bool big = false;
				const bool any_mul = (cd.lflags & (CL_MUL_GOAL | CL_MUL_HOLD)) != 0;
#pragma unroll
				for (uint32_t j = 0; j < 4; ++j) {
					float v = fv[j];
					if (any_mul) v = (th + j < cd.fl.goal_len) ? ((cd.lflags & CL_MUL_GOAL) ? v * cd.mulc : v) : ((cd.lflags & CL_MUL_HOLD) ? v * cd.mulc : v);
					x[j] = cd.coeff * v;
					big |= !(fabsf(x[j]) < 0x1p50f);
				}
				/* llrintf(x) mod 2^32 (wosc.h:145): one test per four frames for the rounding form */
				if (!__any(big)) {
#pragma unroll
					for (uint32_t j = 0; j < 4; ++j) incs[j] = (uint32_t)__double2loint((double)x[j] + 0x1.8p52);
				} else {
#pragma unroll
					for (uint32_t j = 0; j < 4; ++j) incs[j] = rint32w(x[j]);
				}
			} else {
#pragma unroll
				for (uint32_t j = 0; j < 4; ++j) incs[j] = cd.inc_const;
			}
			*(uint4 *)(in_base + chain_io_word(half * HQ + q, l)) = make_uint4(incs[0], incs[1], incs[2], incs[3]);
			*(float4 *)(in_amt + chain_io_word(half * HQ + q, l)) = make_float4(m[0], m[1], m[2], m[3]);
		}
		return;
	}
	if (half) return; /* (rows from HBM: the first feeder wave's alone) */
#pragma unroll
	for (uint32_t q = 0; q < CHAIN_NQ; ++q) {
		uint4 b = bp[q]; /* (fetched a batch ahead: chain_fetch) */
		if (chain_mode(cd) == CM_INC) { /* phase increments: summed here */
			const uint32_t i = t + 4 * q;
			b.x += a; b.y += b.x; b.z += b.y; b.w += b.z;
			a = b.w;
			a_end = i + 3 < cd.n ? b.w : i + 2 < cd.n ? b.z : i + 1 < cd.n ? b.y : i < cd.n ? b.x : a_end;
		}
		*(uint4 *)(in_base + chain_io_word(q, l)) = b;
		*(float4 *)(in_amt + chain_io_word(q, l)) = ap[q];
	}
	*acc = a_end;
}

/* (-DCHAIN_NT_ROWS: streaming accesses for the rows here as well -- measured 61.6 against 51.8 ms per config-5 step, r03:
 * the feeder's loads then always go to HBM, where ordinary ones find part of what the chain-input pass wrote in the caches) */
typedef uint32_t __attribute__((ext_vector_type(4))) chain_u32x4;
typedef float __attribute__((ext_vector_type(4))) chain_f32x4;
__device__ __forceinline__ uint4 chain_ld(const uint4 *p) {
#ifdef CHAIN_NT_ROWS
	const chain_u32x4 v = __builtin_nontemporal_load((const chain_u32x4 *)p);
	return make_uint4(v.x, v.y, v.z, v.w);
#else
	return *p;
#endif
}
__device__ __forceinline__ float4 chain_ld(const float4 *p) {
#ifdef CHAIN_NT_ROWS
	const chain_f32x4 v = __builtin_nontemporal_load((const chain_f32x4 *)p);
	return make_float4(v.x, v.y, v.z, v.w);
#else
	return *p;
#endif
}
__device__ __forceinline__ void chain_st(float4 *p, const float4 v) {
#ifdef CHAIN_NT_ROWS
	chain_f32x4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
	__builtin_nontemporal_store(w, (chain_f32x4 *)p);
#else
	*p = v;
#endif
}
/* the rows' share of batch t into registers: issued one batch before chain_feed needs it, so that the feeder
 * never stands waiting for HBM inside the batch period (1.9 us) the chain wave gives it */
__device__ __forceinline__ void chain_fetch(const ChainDesc &cd, bool live, uint32_t t, const uint4 *brow, const float4 *arow,
		uint4 *bp, float4 *ap) {
	if (!live || chain_mode(cd) == CM_INLINE) return;
#pragma unroll
	for (uint32_t q = 0; q < CHAIN_NQ; ++q) { bp[q] = chain_ld(&brow[t / 4 + q]); ap[q] = chain_ld(&arow[t / 4 + q]); }
}

#ifdef CHAIN_PROF /* tuning aid: where each wave of workgroup 0 spends a launch -- cycles in all, cycles waiting at the barrier */
#define CHAIN_SYNC() do { const uint64_t w0_ = wall_clock64(); __syncthreads(); prof_wait += wall_clock64() - w0_; } while (0)
#define CHAIN_PROF_END() do { if (blockIdx.x == 0 && l == 0) printf("chain_kernel role %u: %llu ticks, %llu at the barrier, %u batches\n", role, (unsigned long long)(wall_clock64() - prof_t0), (unsigned long long)prof_wait, n_batches); } while (0)
#else
#define CHAIN_SYNC() __syncthreads()
#define CHAIN_PROF_END() do {} while (0)
#endif
__global__ void __launch_bounds__(192) chain_kernel(FastParams P) {
	extern __shared__ __align__(16) unsigned char lds[];
	if (P.pass_flags[FAST_MAX_LEVELS + 1] == 0) return; /* no voice of the segment has a chain */
	if (P.chain_early && P.pass_flags[FAST_EARLY_FLAG] == 0) return;
	const int l = threadIdx.x & 63;
	const uint32_t role = uni((uint32_t)threadIdx.x >> 6); /* 0: the chain wave; 1, 2: feeder waves (2: only the second half of inline chains' batches) */
	const bool feeder = role != 0;
	const uint32_t c = blockIdx.x * 64 + (uint32_t)l;
	ChainDesc cd;
	memset(&cd, 0, sizeof cd);
	if (c < P.n_chain_slots && P.chain_desc[c].n != 0) cd = P.chain_desc[c]; /* (an unused lane has only `n` set) */
	/* this launch's share of the chain: frames [c_lo, n) of the segment, n cut at the chunk's end */
	const uint32_t c_lo = P.range_mode ? P.f_lo : 0u;
	uint32_t n = cd.n;
	if (((cd.lflags & CL_EARLY) != 0) != (P.chain_early != 0) || (cd.lflags & CL_RASEG)) n = 0; /* the early chains have a launch of their own, ahead of the passes */
	if (P.range_mode && n > P.f_hi) n = P.f_hi;
	if (n <= c_lo) n = 0;
	if (!__any(n != 0)) return;
	uint32_t *io = (uint32_t *)(lds + (size_t)P.n_ctabs * CHAIN_TAB_BYTES);
	for (uint32_t t = 0; t < P.n_ctabs; ++t) {
		typedef double __attribute__((ext_vector_type(2))) f64x2;
		const uint32_t wave = P.cwave_of_tab[t];
		const uint4 *s23 = (const uint4 *)(P.g_c23 + (size_t)wave * WAVE_LEN);
		uint4 *d23 = (uint4 *)(lds + (size_t)t * CHAIN_TAB_BYTES);
		for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += 192) d23[i] = s23[i];
		const HerpC01 *s01 = P.g_c01 + (size_t)wave * WAVE_LEN;
		f64x2 *d01 = (f64x2 *)(lds + (size_t)t * CHAIN_TAB_BYTES + CHAIN_TAB_C01);
		for (uint32_t i = threadIdx.x; i < WAVE_LEN; i += 192) { f64x2 v; v.x = (double)s01[i].c1; v.y = (double)s01[i].c0; d01[i] = v; }
	}
	const uint32_t wave = chain_wave_id(cd) < 12 ? chain_wave_id(cd) : 0;
	const int ti = P.ctab_of_wave[wave];
	const bool all_lds = __all(n == 0 || ti >= 0) != 0;
	const uint32_t tab23 = (uint32_t)(uintptr_t)lds + (uint32_t)(ti >= 0 ? ti : 0) * CHAIN_TAB_BYTES;
	const uint32_t tab01 = tab23 + CHAIN_TAB_C01;
	const HerpC23 *g23 = P.g_c23 + (size_t)wave * WAVE_LEN;
	const HerpC01 *g01 = P.g_c01 + (size_t)wave * WAVE_LEN;
	const float dscale = P.wc[wave].diff_scale, doff = P.wc[wave].diff_offset;
	DevOp &o = P.ops[cd.gop];
	/* row pair of the chain (idle lanes: pair 0, reads only) */
	float *brow = P.chain_rows + (size_t)2 * (n ? cd.row : 0u) * P.chain_stride;
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
#ifdef CHAIN_PROF
	uint64_t prof_wait = 0; const uint64_t prof_t0 = wall_clock64();
#endif
	if (role == 2) { /* the second feeder wave: lines of inline chains, second half of every batch; the same barriers as the first */
		uint32_t acc = 0;
		for (uint32_t k = 0; k <= n_batches; ++k) {
			if (k < n_batches && c_lo + k * CHAIN_BATCH < n && chain_mode(cd) == CM_INLINE)
				chain_feed(cd, true, l, c_lo + k * CHAIN_BATCH, &acc, nullptr, nullptr, in_base(k & 1), in_amt(k & 1), 1u);
			CHAIN_SYNC();
		}
		CHAIN_PROF_END();
		return;
	}
	if (feeder) {
		uint32_t acc = c_lo ? o.st_phase : o.phase; /* CM_INC: the phase accumulator (staged by the chunk before) */
		/* step k: feed batch k while the chain wave runs batch k - 1, store the samples of batch k - 2; the rows of
		 * batch k + 1 are asked for now and used in the next step */
		uint4 fb[CHAIN_NQ]; float4 fa[CHAIN_NQ];
#pragma unroll
		for (int q = 0; q < (int)CHAIN_NQ; ++q) { fb[q] = make_uint4(0, 0, 0, 0); fa[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
		chain_fetch(cd, n_batches && c_lo < n, c_lo, bp, ap, fb, fa);
		for (uint32_t k = 0; k <= n_batches; ++k) {
			if (k < n_batches && c_lo + k * CHAIN_BATCH < n)
				chain_feed(cd, true, l, c_lo + k * CHAIN_BATCH, &acc, fb, fa, in_base(k & 1), in_amt(k & 1), 0u);
			chain_fetch(cd, k + 1 < n_batches && c_lo + (k + 1) * CHAIN_BATCH < n, c_lo + (k + 1) * CHAIN_BATCH, bp, ap, fb, fa);
			if (k >= 2 && c_lo + (k - 2) * CHAIN_BATCH < n) {
				const float *sq = out_s(k & 1);
#pragma unroll
				for (uint32_t q = 0; q < CHAIN_NQ; ++q) chain_st(&op[(c_lo + (k - 2) * CHAIN_BATCH) / 4 + q], *(const float4 *)(sq + chain_io_word(q, l)));
			}
			CHAIN_SYNC(); /* (the first one also: tables staged) */
		}
		if (n_batches && c_lo + (n_batches - 1) * CHAIN_BATCH < n) {
			const float *sq = out_s((n_batches - 1) & 1);
#pragma unroll
			for (uint32_t q = 0; q < CHAIN_NQ; ++q) chain_st(&op[(c_lo + (n_batches - 1) * CHAIN_BATCH) / 4 + q], *(const float4 *)(sq + chain_io_word(q, l)));
		}
		if (n && chain_mode(cd) == CM_INC) o.st_phase = acc; /* (inline chains: the chain wave's, which sums their increments) */
		CHAIN_PROF_END();
		return;
	}
	/* ---- the chain wave ---- */
	/* the operator's state, or what the chunk before this one staged */
	uint32_t prev_phase = c_lo ? o.st_prev_phase : o.prev_phase;
	double prev_Is = c_lo ? o.st_prev_Is : o.prev_Is;
	float prev_s = c_lo ? o.st_prev_s : o.prev_s, fb_s = c_lo ? bits_f(o.ras_alpha) : o.fb_s;
	const bool inl = chain_mode(cd) == CM_INLINE; /* the feeder waves hand this chain phase increments: summed here */
	const int inl_kind = __all(n == 0 || !inl) ? 0 : __all(n == 0 || inl) ? 1 : 2; /* (per wave: the usual bank is all of one kind) */
	/* (a chain of one frequency stages no accumulator -- finalize_kernel advances its phase in closed form -- so a later
	 * chunk's start is the closed form too) */
	uint32_t acc = (inl && (cd.lflags & CL_FCONST)) ? o.phase + cd.inc_const * c_lo : (c_lo ? o.st_phase : o.phase);
	CHAIN_SYNC();
	if (n && c_lo == 0 && (o.flags & OPF_OSC_RESET)) { /* wosc.h:215-231 with the first base phase, as the block loop does */
		const uint32_t phase00 = (inl ? acc : 0u) + in_base(0)[chain_io_word(0, l)];
		const uint32_t pa = phase00 - SLEN;
		prev_Is = herp_poly(g23[pa >> SLEN_BITS], g01[pa >> SLEN_BITS], pa);
		const double Is0 = herp_poly(g23[phase00 >> SLEN_BITS], g01[phase00 >> SLEN_BITS], phase00);
		prev_s = wosc_reset_s(Is0, herp_poly_rise(g23[pa >> SLEN_BITS], g01[pa >> SLEN_BITS], pa), g01[pa >> SLEN_BITS].c0, dscale, doff);
		prev_Is = Is0;
		prev_phase = phase00;
	}
	for (uint32_t k = 0; k < n_batches; ++k) {
		const uint32_t t = c_lo + k * CHAIN_BATCH;
		uint4 bq[CHAIN_NQ]; float4 aq[CHAIN_NQ]; float4 sq[CHAIN_NQ];
		const uint32_t *ib = in_base(k & 1);
		const float *ia = in_amt(k & 1);
#pragma unroll
		for (uint32_t q = 0; q < CHAIN_NQ; ++q) { bq[q] = *(const uint4 *)(ib + chain_io_word(q, l)); aq[q] = *(const float4 *)(ia + chain_io_word(q, l)); }
		/* the short rounding form needs |fb_s * amount| < 2^20: amounts below 2^14 and |fb_s| <= 64 (checked after) */
		float a_max = 0.f; /* (v_max3_f32 with |.| operands, two per four amounts: fmaxf(fabsf()) chains compiled to seven -- each
		                    * operand canonicalised by a v_max of its own --, 1.25 instructions per sample step of the chain wave) */
#pragma unroll
		for (int u = 0; u < (int)CHAIN_NQ; ++u) {
			asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a_max) : "v"(aq[u].x), "v"(aq[u].y));
			asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a_max) : "v"(aq[u].z), "v"(aq[u].w));
		}
		const uint32_t s_prev_phase = prev_phase, s_acc = acc; const double s_prev_Is = prev_Is;
		const float s_prev_s = prev_s, s_fb_s = fb_s;
		float fb_max = fabsf(fb_s);
		bool small = !__any(n != 0 && !(a_max < 0x1p14f));
		const bool tail = !((k + 1) * CHAIN_BATCH <= n_all);
#define SAU_CHAIN_BATCH_I(L, TL, SM, I) chain_batch<L, TL, SM, I>(bq, aq, sq, t, n, tab23, tab01, g23, g01, dscale, doff, prev_phase, prev_Is, prev_s, fb_s, fb_max, inl, acc)
#define SAU_CHAIN_BATCH(L, TL, SM) do { if (inl_kind == 0) SAU_CHAIN_BATCH_I(L, TL, SM, 0); else if (inl_kind == 1) SAU_CHAIN_BATCH_I(L, TL, SM, 1); else SAU_CHAIN_BATCH_I(L, TL, SM, 2); } while (0)
		if (small) {
			if (all_lds) { if (tail) SAU_CHAIN_BATCH(true, true, true); else SAU_CHAIN_BATCH(true, false, true); }
			else { if (tail) SAU_CHAIN_BATCH(false, true, true); else SAU_CHAIN_BATCH(false, false, true); }
			if (__any(n != 0 && !(fb_max <= 64.f && fabsf(fb_s) <= 64.f))) { /* (never seen: feedback is an average of samples) */
				small = false;
				prev_phase = s_prev_phase; prev_Is = s_prev_Is; prev_s = s_prev_s; fb_s = s_fb_s; acc = s_acc;
			}
		}
		if (!small) {
			if (all_lds) { if (tail) SAU_CHAIN_BATCH(true, true, false); else SAU_CHAIN_BATCH(true, false, false); }
			else { if (tail) SAU_CHAIN_BATCH(false, true, false); else SAU_CHAIN_BATCH(false, false, false); }
		}
#undef SAU_CHAIN_BATCH
#undef SAU_CHAIN_BATCH_I
		float *os = out_s(k & 1);
#pragma unroll
		for (uint32_t q = 0; q < CHAIN_NQ; ++q) *(float4 *)(os + chain_io_word(q, l)) = sq[q];
		CHAIN_SYNC();
	}
	CHAIN_PROF_END();
	if (n) { /* staged: finalize_kernel makes it the operator's state unless the voice's segment is redone */
		o.st_prev_phase = prev_phase;
		o.st_prev_Is = prev_Is;
		o.st_prev_s = prev_s;
		o.ras_alpha = f_bits(fb_s);
		o.ras_level = CHAIN_MARK;
		if (inl && !(cd.lflags & CL_FCONST)) o.st_phase = acc;
	}
}

/* R-oscillator feedback (rasg.h:242-294: per sample, feedback -> phase offset -> cycle carry -> segment value ->
 * feedback), lanes = chains, for R oscillators fed from their own lines (CL_RASEG: always early chains, whole
 * segment, ahead of every pass); samples go to the chain's first row, where every pass that needs them reads them
 * (FT_CHAIN_EARLY). Round 5: what chain_kernel has. A workgroup is two waves over the same 64
 * chains. The CHAIN wave runs nothing but the recurrence -- amount x feedback, the phase offset, the reference build's
 * integer floor, the segment's value, the feedback average -- on inputs it finds in LDS, thirty-two frames per lane at a
 * time, and leaves its samples there. The FEEDER wave does everything that does not depend on the feedback: it evaluates the
 * frequency line and sums the 64-bit counter (post-increment, rasg.h:184-186: what a frame reads is the counter before its
 * own increment), splits it into cycle and phase, evaluates the amount line, a batch ahead; and it writes the batch of
 * samples before to the chain's row, sixteen bytes at a time (until round 5: one lane's 4-byte store per sample, and
 * both line evaluations, on the dependent chain: 480 ns per frame for a single voice). One barrier per batch.
 * Round 6: TWO feeder waves. One wave alone issues a dependent instruction every eight to twelve cycles, and a feeder that evaluates
 * two swept lines, rounds, splits and drains took longer per batch than the recurrence it feeds -- banks of one kind measured 312-326 ns
 * per frame with fixed rate, 371-382 with a swept rate, 496-509 with rate and amount swept (tests/tools/gpu_r_feedback_kinds.py), so a
 * mixed bank ran at its slowest feeder's pace. Now the RATE feeder has the frequency line, the counter and its split, the AMOUNT
 * feeder the amount line and the drain; each is shorter than the chain wave's batch. */
constexpr uint32_t RCHAIN_BATCH = 32;
/* A line's values over a batch of frames for a feeder wave, the shape a constant in the loop: fast_line_value() dispatches on the
 * shape per value -- three to five taken scalar branches, some twenty cycles each for a wave alone on its SIMD, and one value's
 * arithmetic a dependent chain -- where a loop per shape issues the batch's independent values back to back. UNI: the shape is the
 * wave's (chains of one kind); else the lanes' own, value by value as before. */
template <bool UNI>
__device__ __forceinline__ void fast_line_batch(const FastLine &fl, const uint32_t t0, float (&v)[RCHAIN_BATCH]) {
	if (UNI) { /* (the sweeps' lengths are the lanes' own: the whole batch inside every lane's sweep, or the general form below) */
		auto run = [&](auto TYPE) __attribute__((always_inline)) {
			Sweep sw = fl.sw;
			sw.type = decltype(TYPE)::value;
#pragma unroll
			for (uint32_t j = 0; j < RCHAIN_BATCH; ++j) v[j] = sweep_value_inl<true>(sw, t0 + j);
		};
		if (__all(t0 + RCHAIN_BATCH <= fl.goal_len)) {
			switch (fl.sw.type) {
			case LN_cos: run(std::integral_constant<uint32_t, LN_cos>{}); return;
			case LN_lin: run(std::integral_constant<uint32_t, LN_lin>{}); return;
			case LN_sah: run(std::integral_constant<uint32_t, LN_sah>{}); return;
			case LN_xpe: run(std::integral_constant<uint32_t, LN_xpe>{}); return;
			case LN_lge: run(std::integral_constant<uint32_t, LN_lge>{}); return;
			case LN_sqe: run(std::integral_constant<uint32_t, LN_sqe>{}); return;
			case LN_cub: run(std::integral_constant<uint32_t, LN_cub>{}); return;
			case LN_smo: run(std::integral_constant<uint32_t, LN_smo>{}); return;
			default: break; /* (the noise shapes, and what else there may be: value by value) */
			}
		}
	}
	if (__all(t0 >= fl.goal_len)) { /* the line holds */
#pragma unroll
		for (uint32_t j = 0; j < RCHAIN_BATCH; ++j) v[j] = fl.hold;
		return;
	}
#pragma unroll 4
	for (uint32_t j = 0; j < RCHAIN_BATCH; ++j) v[j] = fast_line_value(fl, (int)(t0 + j));
}
constexpr size_t RCHAIN_LDS_BYTES = (size_t)(3 + 1 + 2) * 2 * RCHAIN_BATCH * 64 * 4; /* cycle, phase, amount in; samples out; the counter's increments (64 bits): two batches each */
constexpr uint32_t RCHAIN_THREADS = 192; /* the chain wave, the rate feeder, the amount feeder */
__global__ void __launch_bounds__(RCHAIN_THREADS) rchain_kernel(FastParams P) {
	if (P.pass_flags[FAST_EARLY_FLAG] == 0) return;
	extern __shared__ __align__(16) unsigned char rc_lds[];
	uint32_t *const in_cyc = (uint32_t *)rc_lds;                       /* [2][RCHAIN_BATCH][64] */
	float *const in_ph = (float *)(in_cyc + 2 * RCHAIN_BATCH * 64);
	float *const in_am = in_ph + 2 * RCHAIN_BATCH * 64;
	float *const out_s = in_am + 2 * RCHAIN_BATCH * 64;
	uint32_t *const in_inc = (uint32_t *)(out_s + 2 * RCHAIN_BATCH * 64); /* [2][RCHAIN_BATCH][2][64]: low words, high words */
	const int l = threadIdx.x & 63;
	const uint32_t role = uni((uint32_t)threadIdx.x >> 6); /* 0 the chain wave, 1 the rate feeder, 2 the amount feeder */
	const bool feeder = role != 0;
	const uint32_t c = blockIdx.x * 64 + (uint32_t)l;
	ChainDesc cd = P.chain_desc[c < P.n_chain_slots ? c : 0];
	uint32_t n = 0;
	if (c < P.n_chain_slots) {
		if (cd.n && (cd.lflags & CL_RASEG)) n = cd.n;
	}
	uint32_t n_max = n; /* the workgroup's longest chain decides the number of batches (both waves alike) */
#pragma unroll
	for (int sh = 32; sh >= 1; sh >>= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, sh));
	n_max = uni(n_max);
	if (n_max == 0) return;
	const uint32_t nb = (n_max + RCHAIN_BATCH - 1) / RCHAIN_BATCH;
	DevOp *op = n ? &P.ops[cd.gop] : nullptr;
	const unsigned long long act = __ballot(n != 0);
	const int first = act ? __builtin_ctzll(act) : 0;
	if (feeder) {
		const bool rate2x = n && (op->flags & OPF_RATE2X) != 0;
		const float rcoeff = n ? (rate2x ? cd.coeff * 2 : cd.coeff) : 0.f;
		unsigned long long cp = n ? op->cycle_phase : 0ull;
		const unsigned long long inc_c = n ? (unsigned long long)rint64(rcoeff * op->rt_fconst) : 0ull;
		float *row = P.chain_rows + (size_t)2 * (n ? cd.row : 0u) * P.chain_stride;
		/* the whole of the feeder's work, for line shapes and flags that are the lane's own or -- chains of one kind: a bank of
		 * like voices, one voice -- the wave's (two copies of the loop: in the second every value's shape dispatch is a scalar
		 * branch) */
		if (role == 1) {
			/* the rate feeder: frequency line -> the counter's increment per frame, TWO batches ahead of the chain wave (round 6, later:
			 * summing the counter and splitting it went to the amount feeder, which had the time -- with a swept rate this wave's
			 * 330-340 ns per frame were the workgroup's pace, the recurrence itself 230-250) */
			auto feed = [&](const FastLine &fl, const uint32_t lflags, auto UNI_) {
				constexpr bool UNI = decltype(UNI_)::value;
				auto fill = [&](uint32_t k) { /* the increments of batch k */
					uint32_t *ic = in_inc + (k & 1) * RCHAIN_BATCH * 128;
					const uint32_t t0 = k * RCHAIN_BATCH;
					if (t0 >= n) return;
					if (UNI && (lflags & CL_FCONST)) { /* (the wave's rates are fixed) */
#pragma unroll
						for (uint32_t j = 0; j < RCHAIN_BATCH; ++j) { ic[j * 128 + l] = (uint32_t)inc_c; ic[j * 128 + 64 + l] = (uint32_t)(inc_c >> 32); }
						return;
					}
					float v[RCHAIN_BATCH]; /* (indexed by the loop below: in scratch memory, read back off the recurrence's path; the loops
					                        * unrolled whole to keep it in registers ran the feeders at 640 ns per frame) */
					fast_line_batch<UNI>(fl, t0, v);
#pragma unroll 4
					for (uint32_t j = 0; j < RCHAIN_BATCH; ++j) {
						const uint32_t t = t0 + j;
						unsigned long long inc = inc_c;
						if (!(lflags & CL_FCONST)) {
							float x = v[j];
							if (lflags & (t < fl.goal_len ? CL_MUL_GOAL : CL_MUL_HOLD)) x *= cd.mulc;
							inc = (unsigned long long)rint64(rcoeff * x);
						}
						ic[j * 128 + l] = (uint32_t)inc; ic[j * 128 + 64 + l] = (uint32_t)(inc >> 32);
					}
				};
				fill(0);
				__syncthreads();
				if (nb > 1) fill(1);
				__syncthreads();
				for (uint32_t k = 0; k < nb; ++k) {
					if (k + 2 < nb) fill(k + 2);
					__syncthreads();
				}
			};
			/* line shapes and flags that are the lane's own or -- chains of one kind: a bank of like voices, one voice -- the wave's
			 * (two copies of the loop: in the second every value's shape dispatch is a scalar branch) */
			const uint32_t ft0 = (uint32_t)__builtin_amdgcn_readlane((int)cd.fl.sw.type, first);
			const uint32_t lf0 = (uint32_t)__builtin_amdgcn_readlane((int)cd.lflags, first);
			if (!__any(n != 0 && (cd.fl.sw.type != ft0 || cd.lflags != lf0))) {
				FastLine fl = cd.fl;
				fl.sw.type = ft0;
				feed(fl, lf0, std::true_type{});
			} else {
				feed(cd.fl, cd.lflags, std::false_type{});
			}
			return;
		} else {
			/* the amount feeder: amount line; the counter summed over the rate feeder's increments (post-increment, rasg.h:184-186) and
			 * split into cycle and phase; the samples of the batch before to the chain's row */
			auto feed = [&](const FastLine &pl, auto UNI_) {
				constexpr bool UNI = decltype(UNI_)::value;
				auto fill = [&](uint32_t k) {
					float *am = in_am + (k & 1) * RCHAIN_BATCH * 64;
					uint32_t *cy = in_cyc + (k & 1) * RCHAIN_BATCH * 64;
					float *ph = in_ph + (k & 1) * RCHAIN_BATCH * 64;
					const uint32_t *ic = in_inc + (k & 1) * RCHAIN_BATCH * 128;
					const uint32_t t0 = k * RCHAIN_BATCH;
					if (t0 >= n) return;
					float v[RCHAIN_BATCH];
					fast_line_batch<UNI>(pl, t0, v);
#pragma unroll 4
					for (uint32_t j = 0; j < RCHAIN_BATCH; ++j) {
						const uint32_t t = t0 + j;
						const unsigned long long inc = ((unsigned long long)ic[j * 128 + 64 + l] << 32) | ic[j * 128 + l];
						uint32_t cyc; float phf;
						ras_split(cp, cyc, phf);
						if (t < n) cp += inc;
						cy[j * 64 + l] = cyc; ph[j * 64 + l] = phf;
						am[j * 64 + l] = v[j];
					}
				};
				auto drain = [&](uint32_t k) { /* the samples of batch k to the chain's row */
					const float *os = out_s + (k & 1) * RCHAIN_BATCH * 64;
					const uint32_t t0 = k * RCHAIN_BATCH;
					if (t0 >= n) return;
					if (t0 + RCHAIN_BATCH <= n) {
#pragma unroll
						for (uint32_t q = 0; q < RCHAIN_BATCH / 4; ++q) /* (rows are 256-byte aligned: chain_stride is a multiple of 64) */
							*(float4 *)(row + t0 + 4 * q) = make_float4(os[(4 * q) * 64 + l], os[(4 * q + 1) * 64 + l], os[(4 * q + 2) * 64 + l], os[(4 * q + 3) * 64 + l]);
					} else {
						for (uint32_t j = 0; t0 + j < n; ++j) row[t0 + j] = os[j * 64 + l];
					}
				};
				__syncthreads(); /* the increments of batch 0 */
				fill(0);
				__syncthreads();
				for (uint32_t k = 0; k < nb; ++k) {
					if (k + 1 < nb) fill(k + 1);
					if (k >= 1) drain(k - 1);
					__syncthreads();
				}
				drain(nb - 1);
			};
			const uint32_t pt0 = (uint32_t)__builtin_amdgcn_readlane((int)cd.pl.sw.type, first);
			if (!__any(n != 0 && cd.pl.sw.type != pt0)) {
				FastLine pl = cd.pl;
				pl.sw.type = pt0;
				feed(pl, std::true_type{});
			} else {
				feed(cd.pl, std::false_type{});
			}
		}
		if (n) { /* staged: finalize_kernel makes it the operator's state unless the voice's segment is redone */
			op->st_prev_Is = __longlong_as_double((long long)cp);
			op->st_phase = CHAIN_MARK;
		}
		return;
	}
	/* the chain wave */
	RasParams rp;
	float fb_s = 0.f, prev_s = 0.f;
	if (n) {
		rp = ras_params(op->ras_func, op->ras_flags, op->ras_level, op->ras_alpha, op->wave);
		fb_s = op->fb_s; prev_s = op->prev_s;
	} else {
		rp = ras_params(0, 0, 0, 0, 0);
	}
	auto chain = [&](const RasParams &rq) __attribute__((always_inline)) {
		uint32_t e_cyc = 0; float e_a, e_b;
		ras_ends(rq, e_cyc, e_a, e_b); /* (the ends of cycle 0 to begin with: whatever the first sample's cycle, the cache is true) */
		__syncthreads(); /* (the feeders' first hand-over: batch 0's increments) */
		__syncthreads(); /* batch 0's inputs */
		for (uint32_t k = 0; k < nb; ++k) {
			const uint32_t *cy = in_cyc + (k & 1) * RCHAIN_BATCH * 64;
			const float *ph = in_ph + (k & 1) * RCHAIN_BATCH * 64, *am = in_am + (k & 1) * RCHAIN_BATCH * 64;
			float *os = out_s + (k & 1) * RCHAIN_BATCH * 64;
			const uint32_t t0 = k * RCHAIN_BATCH;
			if (t0 < n) {
				const uint32_t m = min(RCHAIN_BATCH, n - t0);
				auto step = [&](const uint32_t cyc_in, const float ph_in, const float am_in, const uint32_t j) {
					const float pm_a = ras_fb_amount(fb_s, am_in);
					float phase = ph_in + pm_a;
					const int32_t cycle_adj = floor_i32_ref(phase); /* (a feedback offset of 2^31 cycles and more: the host's conversion and its wrap) */
					const uint32_t cycle = cyc_in + (uint32_t)cycle_adj;
					phase -= (float)cycle_adj;
					/* (the segment's ends are a function of the cycle: kept while every lane's cycle stands -- any lane in a new cycle and
					 * all recompute, to the same values where nothing moved; two hashes a sample otherwise, two Gauss transforms for the
					 * Gauss function: 317-331 ns per frame there against 243-248 for the others) */
					if (__any(cycle != e_cyc)) { ras_ends(rq, cycle, e_a, e_b); e_cyc = cycle; }
					const float sv = ras_sample_ends(rq, e_a, e_b, phase);
					os[j * 64 + l] = sv;
					fb_s = ((fb_s + prev_s) + sv) * 0.5f; /* the reference build's association (see the oracle) */
					prev_s = sv;
				};
				if (m == RCHAIN_BATCH) {
					/* four steps' inputs into registers ahead of the four steps: the LDS reads go out together and off the chain
					 * (between the steps stand the scalar branches of the function / flag / shape dispatch, which loads do not cross) */
					for (uint32_t j0 = 0; j0 < RCHAIN_BATCH; j0 += 4) {
						uint32_t c4[4]; float p4[4], a4[4];
#pragma unroll
						for (uint32_t u = 0; u < 4; ++u) { c4[u] = cy[(j0 + u) * 64 + l]; p4[u] = ph[(j0 + u) * 64 + l]; a4[u] = am[(j0 + u) * 64 + l]; }
#pragma unroll
						for (uint32_t u = 0; u < 4; ++u) step(c4[u], p4[u], a4[u], j0 + u);
					}
				} else {
					for (uint32_t j = 0; j < m; ++j) step(cy[j * 64 + l], ph[j * 64 + l], am[j * 64 + l], j);
				}
			}
			__syncthreads();
		}
	};
	{ /* (as the feeder's line shapes: function, flags, level and line of a bank of like voices are the wave's, and every
	   * sample's dispatch on them a scalar branch) */
		const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)rp.func, first), g0 = (uint32_t)__builtin_amdgcn_readlane((int)rp.flags, first);
		const uint32_t v0 = (uint32_t)__builtin_amdgcn_readlane((int)rp.level, first), l0 = (uint32_t)__builtin_amdgcn_readlane((int)rp.line, first);
		if (!__any(n != 0 && (rp.func != f0 || rp.flags != g0 || rp.level != v0 || rp.line != l0))) {
			RasParams ru = rp;
			ru.flags = g0; ru.level = v0; ru.line = l0;
			/* ... and a copy of the loop per function (round 6: the loop is inlined with the function a constant, so that ras_ends'
			 * dispatch -- four to six taken scalar branches a sample, some twenty cycles each for a wave alone on its SIMD --
			 * is compiled out; flags and line shape stay scalar branches) */
			switch (f0) {
			case RF_URAND: ru.func = RF_URAND; chain(ru); break;
			case RF_GAUSS: ru.func = RF_GAUSS; chain(ru); break;
			case RF_BIN: ru.func = RF_BIN; chain(ru); break;
			case RF_TERN: ru.func = RF_TERN; chain(ru); break;
			case RF_FIXED: ru.func = RF_FIXED; chain(ru); break;
			case RF_ADDREC: ru.func = RF_ADDREC; chain(ru); break;
			default: ru.func = f0; chain(ru); break;
			}
		} else {
			chain(rp);
		}
	}
	if (n) {
		op->st_prev_s = prev_s;
		op->st_prev_phase = f_bits(fb_s);
	}
}
