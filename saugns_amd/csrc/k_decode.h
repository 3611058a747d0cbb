/* k_decode.h -- part of hip_backend.hip (included there, inside namespace sauhip; not a header of its own).
 * Decoded steps (FastStep / FastLine / FastAux / ChainDesc), scan_kernel and decode_kernel. */
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
enum : uint32_t { CL_FCONST = 1, CL_MUL_GOAL = 2, CL_MUL_HOLD = 4, CL_EARLY = 8 /* runs in the launch ahead of the passes (FastInfo.early) */,
	CL_RASEG = 16 /* an R oscillator: rchain_kernel's (always early) */ };
struct ChainDesc {
	uint32_t n;      /* frames to run this segment (0: row pair unused; the other fields are then unset) */
	uint32_t gop;    /* the operator's state (global index) */
	uint32_t wave_mode; /* the wave table's id | CM_* << 8 */
	uint32_t row;    /* its row pair (entries are numbered by lane: VoiceDesc.chain_slot) */
	float coeff;     /* CM_INLINE: 2^32 / srate */
	uint32_t inc_const; /* ... the phase increment when the frequency is one value (CL_FCONST) */
	uint32_t lflags; /* CL_* */
	float mulc;      /* ... multiplier of a ratio line (the parent's frequency, one value) */
	FastLine fl;     /* ... frequency line over the segment */
	FastLine pl;     /* ... self-modulation amount line */
};
static_assert(sizeof(ChainDesc) == 4 * CHAIN_DESC_WORDS && offsetof(ChainDesc, n) == 0, "ChainDesc is 32 dwords, n first");
__device__ __forceinline__ uint32_t chain_wave_id(const ChainDesc &c) { return c.wave_mode & 0xffu; }
__device__ __forceinline__ uint32_t chain_mode(const ChainDesc &c) { return c.wave_mode >> 8; }

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
	const bool lean = P.lean_on && P.info[v].seq == 2 && P.info[v].n_scan == 0; /* fast_kernel<T, 3>'s */
	const uint32_t NP = 64 * (P.info[v].cub ? FAST_CUB_ROWS : lean ? P.rows_lean : (P.info[v].seq == 1 || P.info[v].seq == 2) ? P.rows_multi :
		(P.split_cf && P.info[v].seq == 0) ? P.rows_cf : P.rows);
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
			if (P.info[v].cub && (vd.flags & VD_TAILS) && o.wave == LN_cub && !o.rt_frozen) {
				/* the reference's loop tails of a `cub` map (sau_dev_math.h: TailCtx): the block that holds a frame ends
				 * where this operator, one it is nested in or the voice stops -- frames from the segment's start, as
				 * the block loop keeps them per nesting level (cur_rem). Backwards over the plan: a BEGIN whose END has
				 * not been met on the way is this operator's or an ancestor's. */
				uint32_t rem = (vd.flags & VD_MORE) ? TAIL_FAR : min(vd.run_len, TAIL_FAR);
				uint32_t closed = 0;
				for (uint32_t q = (uint32_t)l + 1; q-- > 0;) {
					const Step sq = plan[q];
					if (q != (uint32_t)l && (sq.flags & SF_END)) ++closed;
					if (sq.flags & SF_BEGIN) {
						if (closed) --closed;
						else {
							const DevOp &oa = P.ops[ids[sq.op]];
							if (!(oa.flags & OPF_TIME_INF) && oa.time < rem) rem = oa.time;
						}
					}
				}
				f.type |= FT_CUBTAIL;
				f.phase0 = rem;
			}
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
			cd.n = P.info[v].total; cd.gop = ids[st.op]; cd.row = row;
			uint32_t cmode = step_is_chain_acc(st, o) ? CM_INC : CM_BASE;
			if (st.sm == NO_SLOT) { /* the amounts come from the line itself */
				LineState pls = o.line[L_PMA];
				const LineBlock lb = line_begin(pls, P.info[v].total, false, 0.f, lattice_none(), 0);
				pl.sw = lb.sw; pl.goal_len = lb.goal_len; pl.hold = lb.hold; pl.pad = 0;
				chain_line = true;
			}
			uint32_t lstep = ~0u;
			if (step_is_chain_inline(P.chain_inline != 0 || P.info[v].early != 0, plan, (uint32_t)l, ids, P.ops, &lstep)) {
				chain_inline = true;
				cmode = CM_INLINE;
				if (P.info[v].early) { cd.lflags |= CL_EARLY; f.type |= FT_CHAIN_EARLY; }
				if (o.type == OT_RASEG) cd.lflags |= CL_RASEG;
				cd.coeff = o.coeff;
				cd.pl = pl;
				cd.mulc = 1.f;
				if (o.rt_fconst_valid) {
					cd.lflags |= CL_FCONST;
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
			cd.wave_mode = (wv & 0xffu) | (cmode << 8);
			P.chain_desc[vd.chain_slot + k] = cd;
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
				held = wosc_reset_s(Is0, herp_poly_rise(g23[pb >> SLEN_BITS], g01[pb >> SLEN_BITS], pb), g01[pb >> SLEN_BITS].c0,
						P.wc[wv].diff_scale, P.wc[wv].diff_offset);
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
					const bool v_early = P.info[v].early != 0;
					const bool q_chain = step_is_chain(sq, oq);
					uint32_t q_ls = ~0u;
					const bool q_inline = q_chain && step_is_chain_inline(P.chain_inline != 0 || v_early, plan, q, ids, P.ops, &q_ls);
					const bool q_early = q_inline && v_early;
					{
						bool needed = false;
						if (q_inline) {
							/* its inputs are its own lines: the feeder wave of chain_kernel evaluates them */
							if (v_early && writes && (want_c & bit(cq.out))) { /* an early chain another chain's inputs read: from its row */
								needed = true;
								if (!rmw) want_c &= ~bit(cq.out);
								want_c |= bit(cq.amp);
							}
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
						if (q_early) { /* an early chain: its samples are in its row, only the amplitude is computed */
							if (writes && (want[p] & bit(cq.out))) {
								needed = true;
								if (!rmw) want[p] &= ~bit(cq.out);
								want[p] |= bit(cq.amp);
							}
						} else if (q_fvar && q_level == p + 1) {
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
			const bool is_red = o.type == OT_NOISE && o.wave == NZ_re;
			if (seq == 3 && st.kind == ST_OSC && ((is_osc && !o.rt_fconst_valid) || is_red)) {
				/* single-pass voice: which of its look-back arrays this oscillator (or red noise: a running sum too) has */
				uint32_t xi = 0;
				for (uint32_t q = 0; q < (uint32_t)l; ++q) {
					const Step sq = plan[q];
					const DevOp &oq = P.ops[ids[sq.op]];
					if (oq.rt_frozen) continue;
					if (sq.kind == ST_OSC && (((oq.type == OT_WAVE || oq.type == OT_RASEG) && !oq.rt_fconst_valid) ||
					                          (oq.type == OT_NOISE && oq.wave == NZ_re))) ++xi;
				}
				fa.pad[0] = xi; fa.pad[1] = 0;
				if (is_red) f.ramp |= 2; /* (its aux record carries the array's index) */
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
	/* The host numbers a running-sum voice's block buffers without those of frequency lines it takes to be one value for every
	 * segment (sau_dev_types.h: fast_slot_compact, the lean form). Should this segment's state say otherwise -- a line kept
	 * here that has no buffer, a frequency or a ratio's multiplier read from one that does not exist -- the voice is the block
	 * loop's for this segment: slow, never wrong. (The host's rule is analyze_kernel's without the ramps, which it sees coming
	 * as events; SAU_AMD_LEAN_IDS_LIE makes it forget them, for the test of this path.) */
	{
		bool lost = false;
		if ((uint32_t)l < vd.plan_len && seq) {
			const Step st = P.steps[vd.plan_ofs + l];
			const FastIds cs = P.fast_ids[P.ids_full_ofs + vd.plan_ofs + l];
			const DevOp &o = P.ops[ids[st.op]];
			if (keep && st.kind == ST_LINE && st.which == L_FREQ && cs.out == NO_SLOT) lost = true;
			/* the end of a range without a buffer: one value, folded into the blend that reads it (below) -- or not one value after all */
			if (keep && st.kind == ST_LINE && (st.which == L_FREQ2 || st.which == L_AMP2) && cs.out == NO_SLOT)
				keep = false; /* (whether it is one value: its blend's lane, next) */
			if (keep && st.kind == ST_LERP && st.freq != NO_SLOT && cs.aux == NO_SLOT) {
				/* its range end: the line step that filled plan buffer st.freq last */
				bool found = false;
				for (uint32_t q = (uint32_t)l; q-- > 0;) {
					const Step sq = P.steps[vd.plan_ofs + q];
					if (sq.out != st.freq) continue;
					if (sq.kind == ST_LINE && (sq.which == L_FREQ2 || sq.which == L_AMP2)) {
						const DevOp &oq = P.ops[ids[sq.op]];
						const LineState &ls = oq.line[sq.which];
						float c = ls.v0;
						bool pc = true;
						if (sq.fmul != NO_SLOT && (ls.flags & LP_STATE_RATIO)) { /* sau/line.c:72, as the line step would: v0 x the parent's one value */
							pc = sq.prov != NO_SLOT && P.ops[ids[sq.prov]].rt_fconst_valid != 0;
							if (pc) c *= P.ops[ids[sq.prov]].rt_fconst;
							else {
								/* ... or x the parent's block as it stands when the line step reads it (analyze_kernel: rt_fblk_valid then): one
								 * value while the parent's line is held, its own multiplier -- if it is a ratio -- one value, and nothing has been
								 * added into the block yet (a range modulator's rate under a carrier whose own range blend comes later) */
								for (uint32_t r = q; r-- > 0;) {
									const Step sr = P.steps[vd.plan_ofs + r];
									if (sr.out != sq.fmul) continue;
									if (sr.kind == ST_LINE && sr.which == L_FREQ) {
										const DevOp &po = P.ops[ids[sr.op]];
										const LineState &pls = po.line[L_FREQ];
										pc = !(pls.flags & LP_GOAL) && (sr.fmul == NO_SLOT || !(pls.flags & LP_STATE_RATIO) ||
										                                (sr.prov != NO_SLOT && P.ops[ids[sr.prov]].rt_fconst_valid != 0));
										if (pc) c *= po.rt_fconst;
									}
									break; /* (anything else that wrote it: not one value) */
								}
							}
						}
						found = pc && !(ls.flags & LP_GOAL);
						f.fc = c;
					}
					break;
				}
				if (!found) lost = true;
			}
			if (keep && st.kind == ST_OSC && (o.type == OT_WAVE || o.type == OT_RASEG) && !o.rt_fconst_valid && st.freq != NO_SLOT && cs.freq == NO_SLOT) lost = true;
			if (keep && (f.ramp & 2) && fa.fmul_off == ~0u && (fa.flags & (FA_MUL_GOAL | FA_MUL_HOLD)) && fa.mulc == 1.f && st.fmul != NO_SLOT && cs.fmul == NO_SLOT) {
				/* (a multiplier wanted from a block: fine only when the parent's frequency was found to be one value) */
				bool pc = false;
				if (st.prov != NO_SLOT) pc = P.ops[ids[st.prov]].rt_fconst_valid != 0;
				if (!pc) lost = true;
			}
		}
		if (__any(lost)) {
			if (l == 0) { P.info[v].total = 0; P.info[v].bail = 1; }
			keep = false;
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
