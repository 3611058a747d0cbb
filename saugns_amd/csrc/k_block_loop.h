/* k_block_loop.h -- part of hip_backend.hip (included there, inside namespace sauhip; not a header of its own).
 * render_kernel<W,T,V>: the general block loop -- one team of waves per voice, operator state
 * and block buffers in LDS, the reference's block structure kept. */
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

/* the sample a (re)started oscillator begins with (sau_dev_math.h: wosc_reset_s), from the tables */
__device__ __forceinline__ float wosc_reset_lookup(const TabRef &t, uint32_t phase0, double Is0, float diff_scale, float diff_offset) {
	const uint32_t pp = phase0 - SLEN, ind = pp >> SLEN_BITS;
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
	return wosc_reset_s(Is0, herp_poly_rise(hi, lo, pp), lo.c0, diff_scale, diff_offset);
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
template <bool LDS, int SPAN /* samples per wave span when several waves share a block, else 0 */,
          bool SLDS = true /* the block buffers lie in LDS (false: in HBM, render_kernel<.., HB>: plain pointers) */>
__device__ __forceinline__ void selfmod_serial(const TabRef &tab, SelfmodState &st, const WaveConst &wc,
		const float *pmaS, u32_alias *baseS /* also receives the samples */, uint32_t len) {
	typedef uint32_t __attribute__((address_space(3))) *lds_u32_w_;
	using lds_u32_w = std::conditional_t<SLDS, lds_u32_w_, uint32_t *>;
	using lds_f32_ptr = std::conditional_t<SLDS, ::sauhip::lds_f32_ptr, const float *>;
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

/* HB: the block buffers of this workgroup's voice lie in HBM (P.big_slots, one area per workgroup) instead
 * of LDS -- voices with more buffers than LDS holds (wide plans of very deep graphs: one wave, one frame per
 * lane, 256 B per buffer; a 256-level chain has 500-800 of them beside 64 KiB of operator states). */
template <int W, int T, int V, bool HB = false>
__global__ void __launch_bounds__(64 * W * V) render_kernel(RenderParams P) {
	static_assert(V == 1 || W == 1, "several teams per workgroup are single waves");
	static_assert(!HB || (W == 1 && V == 1), "buffers in HBM: one wave per workgroup");
	using G = Geo<W, T>;
	constexpr int NTHREADS = 64 * W * V;
	extern __shared__ __align__(16) unsigned char lds[];
	const int team = V > 1 ? (int)uni((uint32_t)threadIdx.x >> 6) : 0;
	const int tid = V > 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x; /* within the team */
	const int w = tid >> 6;
	const int l = tid & 63;

	HerpC23 *t23 = (HerpC23 *)lds;
	HerpC01 *t01 = (HerpC01 *)(lds + (size_t)P.n_tabs * WAVE_LEN * sizeof(HerpC23));
	unsigned char *team_lds = lds + (size_t)P.n_tabs * WAVE_LEN * (sizeof(HerpC23) + sizeof(HerpC01)) +
			(size_t)team * P.team_bytes;
	float *slots;
	DevOp *ops;
	Misc *misc;
	Step *plan; /* this voice's steps, read every block */
	if constexpr (HB) {
		/* block buffers, operator records and steps in the workgroup's area in HBM (what bounds a voice then is memory,
		 * not LDS: thousands of operators), the bookkeeping in LDS */
		slots = P.big_slots + (size_t)blockIdx.x * P.big_stride;
		ops = (DevOp *)(slots + (size_t)P.n_slots * G::SLOT);
		plan = (Step *)(ops + P.max_ops);
		misc = (Misc *)team_lds;
	} else {
		slots = (float *)team_lds;
		ops = (DevOp *)(slots + (size_t)P.n_slots * G::SLOT);
		misc = (Misc *)(ops + P.max_ops);
		plan = (Step *)(misc + 1);
	}

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
		/* frames until the operator being evaluated, an ancestor or the voice stops: where the reference cuts its
		 * blocks (the loop tails of `cub`, sau_dev_math.h: TailCtx) */
		uint32_t cur_rem = (vd.flags & VD_MORE) ? TAIL_FAR : min(vd.run_len - done, TAIL_FAR);
		bool block_ended = false;

		for (uint32_t si = 0; si < vd.plan_len && !block_ended; ++si) {
			/* buffer ids -> memory indices (two pools; wide plans: the next step holds the ids' high bytes:
			 * sau_dev_types.h) */
			const Step st_lo = uni(plan[si]);
			WideStep st;
			if (vd.flags & VD_WIDE) {
				const Step st_hi = uni(plan[si + 1]);
				++si;
				st = step_widen(st_lo, &st_hi, P.n_main);
			} else {
				st = step_widen(st_lo, nullptr, P.n_main);
			}
			const uint32_t parent_len = cur_len, parent_rem = cur_rem;
			DevOp *op = &ops[st.op];
			const uint32_t op_flags = uni(op->flags);
			if (st.flags & SF_BEGIN) { /* generator.c:694-698 */
				if (tid == 0) { misc->len_stack[depth] = (uint16_t)cur_len; misc->rem_stack[depth] = (uint16_t)cur_rem; }
				++depth;
				const uint32_t op_time = uni(op->time);
				if (!(op_flags & OPF_TIME_INF) && op_time < cur_len) cur_len = op_time;
				if (!(op_flags & OPF_TIME_INF) && op_time < cur_rem) cur_rem = op_time;
			}
			const uint32_t len = cur_len;
			TailCtx tc;
			tc.lat = lat; tc.ev_left = uni(vd.ev_left); tc.off = done; tc.rem = cur_rem; tc.on = (vd.flags & VD_TAILS) ? 1u : 0u;
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
				const float *mul = st.fmul != NO_WSLOT ? slots + (size_t)st.fmul * G::SLOT : nullptr;
				LineState ls = uni(op->line[st.which]);
				/* a held frequency is passed on as one value instead of a block */
				bool pconst = false; float pf = 0.f;
				if (mul && st.prov != NO_WSLOT) { pconst = uni(ops[st.prov].rt_fconst_valid) != 0; pf = uni(ops[st.prov].rt_fconst); }
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
							v[k] = ((tc.on && lb.sw.type == LN_cub) ? line_value_vt(lb, (uint32_t)(jbase + k), m, tc) : line_value_v(lb, (uint32_t)(jbase + k), m));
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
						v[k] = owned[k] ? ((tc.on && lb.sw.type == LN_cub) ? line_value_vt(lb, (uint32_t)(jbase + k), 1.f, tc) : line_value_v(lb, (uint32_t)(jbase + k), 1.f)) : 0.f;
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
				const float *fslot = st.freq != NO_WSLOT ? slots + (size_t)st.freq * G::SLOT : nullptr;
				const float *fmul = st.fmul != NO_WSLOT ? slots + (size_t)st.fmul * G::SLOT : nullptr;
				const float *pmS = st.pm != NO_WSLOT ? slots + (size_t)st.pm * G::SLOT : nullptr;
				const float *fpmS = st.fpm != NO_WSLOT ? slots + (size_t)st.fpm * G::SLOT : nullptr;
				const float *ampS = st.amp != NO_WSLOT ? slots + (size_t)st.amp * G::SLOT : nullptr;
				const float *smS = st.sm != NO_WSLOT ? slots + (size_t)st.sm * G::SLOT : nullptr;
				const uint32_t type = uni(op->type);
				const bool is_osc = (type == OT_WAVE || type == OT_RASEG);
				const bool wave_env = (st.flags & SF_WAVE_ENV) != 0;
				const bool layer = (st.flags & SF_LAYER) != 0;

				float s[T], av[T], dv[T];
#pragma unroll
				for (int k = 0; k < T; ++k) { s[k] = 0.f; av[k] = 0.f; dv[k] = 0.f; }

				/* ---- frequency: one value for the block, a slot, or a line ---- */
				bool pconst = false; float pf = 0.f;
				if (st.prov != NO_WSLOT) { pconst = uni(ops[st.prov].rt_fconst_valid) != 0; pf = uni(ops[st.prov].rt_fconst); }
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
							av[k] = owned[k] ? ((tc.on && alb.sw.type == LN_cub) ? line_value_vt(alb, (uint32_t)(jbase + k), 1.f, tc) : line_value_v(alb, (uint32_t)(jbase + k), 1.f)) : 0.f;
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
								                : ((tc.on && flb.sw.type == LN_cub) ? line_value_vt(flb, (uint32_t)j, fmul ? (mconst ? pf : fmul[e]) : 1.f, tc) : line_value_v(flb, (uint32_t)j, fmul ? (mconst ? pf : fmul[e]) : 1.f));
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
								if (owned[k]) out[w * G::NP + p0 + k] = ((tc.on && plb.sw.type == LN_cub) ? line_value_vt(plb, (uint32_t)(jbase + k), 1.f, tc) : line_value_v(plb, (uint32_t)(jbase + k), 1.f));
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
								prev_s = wosc_reset_lookup(tab, phase00, Is0, wc.diff_scale, wc.diff_offset);
								prev_Is = Is0;
								prev_phase = phase00;
							}
							SelfmodState ss;
							ss.prev_phase = prev_phase; ss.prev_Is = prev_Is; ss.prev_s = prev_s; ss.fb_s = fb_s;
							constexpr int SPAN = W > 1 ? G::NP - 1 : 0;
							if (tab.in_lds) selfmod_serial<true, SPAN, !HB>(tab, ss, wc, pmaS, scratch_u, len);
							else selfmod_serial<false, SPAN, !HB>(tab, ss, wc, pmaS, scratch_u, len);
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
							                : ((tc.on && flb.sw.type == LN_cub) ? line_value_vt(flb, (uint32_t)j, fmul ? (mconst ? pf : fmul[e]) : 1.f, tc) : line_value_v(flb, (uint32_t)j, fmul ? (mconst ? pf : fmul[e]) : 1.f)));
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
							if (owned[k]) s[k] = ras_sample(rp, cyc[k], phf[k], true, rp.line == LN_cub && cub_map_is_tail(tc, (uint32_t)(jbase + k))); /* rasg.h:692-743 */
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
								float pma_v = smS ? smS[e] : ((tc.on && plb.sw.type == LN_cub) ? line_value_vt(plb, j, 1.f, tc) : line_value_v(plb, j, 1.f));
								float pm_a = ras_fb_amount(fb_s, pma_v);
								float phase = scratch[e] + pm_a;
								int32_t cycle_adj = floor_i32_ref(phase); /* (a feedback offset of 2^31 cycles and more: the host's conversion and its wrap) */
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
							prev_s = wosc_reset_lookup(tab, phase00, Is0, wc.diff_scale, wc.diff_offset);
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
							if (prow) prow[done + j] = pan_goal ? ((tc.on && plb2.sw.type == LN_cub) ? line_value_vt(plb2, (uint32_t)j, 1.f, tc) : line_value_v(plb2, (uint32_t)j, 1.f)) : pl.v0;
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
				const float *panS = st.pm != NO_WSLOT ? slots + (size_t)st.pm * G::SLOT : nullptr;
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
								: (pan_goal ? ((tc.on && plb2.sw.type == LN_cub) ? line_value_vt(plb2, (uint32_t)j, 1.f, tc) : line_value_v(plb2, (uint32_t)j, 1.f)) : pl.v0);
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
				const uint32_t outer_rem = (st.flags & SF_BEGIN) ? parent_rem : uni((uint32_t)misc->rem_stack[depth]);
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
					cur_rem = outer_rem;
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
