/* k_finish.h -- part of hip_backend.hip (included there, inside namespace sauhip; not a header of its own).
 * finalize_kernel, premix_kernel, mix_kernel, mix_few_kernel, event_kernel. */
/* Apply the closed forms to the operator state, or hand the whole segment
 * to the block loop when a chunk had to bail out. */
__global__ void __launch_bounds__(64) finalize_kernel(FastParams P) {
	/* eight threads per (voice, operator): one per line (a held line's walk over the reference's block lattice
	 * is a loop over the host's calls inside the segment -- the long pole of this kernel), one for the rest of the
	 * operator's state; the voice's own bookkeeping goes to that thread of its operator 0 */
	const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
	const uint32_t sub = gid & 7u, vo_ = gid >> 3;
	static_assert(L_COUNT <= 6, "threads 0..5 of an operator's eight take its lines, thread 6 the rest");
	const uint32_t v = vo_ / P.max_ops, i = vo_ % P.max_ops;
	if (gid < FAST_FLAGS && P.pass_flags) P.pass_flags[gid] = 0; /* for the next segment's kernels */
	if (v >= P.n_voices) return;
	const FastInfo fi = P.info[v];
	const VoiceDesc vd = P.voices[v];
	if (fi.total == 0 || fi.bail) {
		if (i == 0 && sub == 6) {
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
		const bool o_osc = o.type == OT_WAVE || o.type == OT_RASEG;
		const Step *plan = P.steps + vd.plan_ofs;
		for (uint32_t ln = sub; ln < L_COUNT; ln += 8) { /* (this thread's line, if it has one) */
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
		if (sub == 6) {
		if (!(o.flags & OPF_TIME_INF)) o.time -= total;
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
			if (o.st_phase == CHAIN_MARK) { /* a feedback chain: rchain_kernel staged the counter and the feedback state */
				o.cycle_phase = (unsigned long long)__double_as_longlong(o.st_prev_Is);
				o.prev_s = o.st_prev_s;
				o.fb_s = bits_f(o.st_prev_phase);
				o.st_phase = 0;
			}
			else if (o.rt_fconst_valid) o.cycle_phase += inc64 * total;
			else o.cycle_phase = (unsigned long long)__double_as_longlong(o.st_prev_Is); /* running sum, staged */
		} else if (o.type == OT_NOISE) {
			const uint32_t n0 = o.noise_n;
			if (o.wave == NZ_vi) o.noise_prev = ranfast32(n0 + total - 1);
			else if (o.wave == NZ_bv) o.noise_prev = (uint32_t)noise_bv_term(n0 + total - 1);
			else if (o.wave == NZ_re) o.noise_prev = o.st_prev_phase; /* the running sum, staged by fast_voice */
			o.noise_n = n0 + total;
		}
		}
		}
	}
	if (i != 0 || sub != 6) return;
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

/* Segments with feedback chains run in chunks of frames, chains beside passes (hip_backend.hip), and the mixer -- 1.07 ms of
 * reading 7 GB of voice rows for BASELINE config 5's 10 s -- used to wait for the last of them. It can take a chunk's frames
 * as soon as the chunk's final pass has written them, provided it knows what finalize_kernel will tell it: this kernel writes
 * every voice's mixer record ahead of the passes -- the constant pan, the frames the time-parallel path will render -- and
 * notes in work_count[1] when some voice is not one it can speak for (not on that path, or not to its end: the block loop will
 * rewrite its row). The early launches then mix frame windows; the launch after finalize_kernel mixes what is left -- the
 * whole segment, once more, when the note or a voice on the block loop's work list (a bail-out found in a pass) says the
 * early results do not stand. Sums and order are the mixer's own: the same PCM either way (round 5). */
__global__ void __launch_bounds__(64) premix_kernel(FastParams P) {
	const uint32_t v = blockIdx.x * 64 + threadIdx.x;
	if (P.inmix_flags & 32u) { /* ahead of a closed-form launch that mixes (k_fast_types.h): its control words */
		/* the chunks' frame ranges, as fast_voice cuts voice 0's (and, checked below, every voice's) row groups into dyn_chunks runs */
		const FastInfo f0 = P.info[0];
		const uint32_t GF = 64u * P.rows - f0.H;
		const uint32_t ngroups = f0.total && f0.H < 64u * P.rows ? (f0.total + GF - 1) / GF : 0u;
		const uint32_t K = P.dyn_chunks ? P.dyn_chunks : 1u;
		uint32_t lo_, hi_;
		const FkChunks q = fk_chunk_groups(ngroups, K, P.dyn_small, 0u, lo_, hi_);
		const uint32_t per = q.per;
		const bool tapered = q.body < ngroups; /* (then the last INMIX_NSMALL chunks are P.dyn_small groups each) */
		const uint32_t nch = q.nch1 + (tapered ? INMIX_NSMALL : 0u);
		const uint32_t tpc = (per * GF + INMIX_TILE - 1) / INMIX_TILE;
		const bool fits = nch != 0 && nch <= INMIX_MAX_CHUNKS && tpc <= INMIX_MAX_TPC && (per * GF) % 4u == 0 && /* (16-byte row loads: chunks begin on a multiple of four frames) */
			(!tapered || ((q.body * GF) % 4u == 0 && (P.dyn_small * GF) % 4u == 0));
		if (v == 0) {
			P.inmix[INMIX_CF] = per * GF;
			P.inmix[INMIX_TPC] = tpc;
			P.inmix[INMIX_NCH1] = q.nch1;
			P.inmix[INMIX_BASE] = tapered ? q.body * GF : ~0u;
			P.inmix[INMIX_CFS] = tapered ? P.dyn_small * GF : 0u;
			P.inmix[INMIX_NCH] = fits ? nch : 0u;
			if (!fits) atomicOr(&P.work_count[1], 4u);
		}
		/* the queues' counters; per chunk the count and the tiles' bits */
		const uint32_t lines = 9 + (fits ? nch : 0u);
		for (uint32_t i = v; i < lines * 9; i += gridDim.x * 64) {
			const uint32_t ln = i / 9, wd = i % 9;
			if (ln) P.inmix[ln * INMIX_LINE + (wd ? INMIX_BITS + wd - 1 : 0u)] = 0;
		}
	}
	if (v >= P.n_voices) return;
	const FastInfo fi = P.info[v];
	const VoiceDesc vd = P.voices[v];
	if (P.inmix_flags & 32u) {
		const FastInfo f0 = P.info[0];
		if (fi.H != f0.H || fi.total != f0.total || vd.pan_dynamic_row != ~0u || fi.total < P.inmix_stream->write_len || fi.cub)
			atomicOr(&P.work_count[1], 4u);
	}
	/* (seq == 1: a voice one wave walks in order -- more than eight running sums, or look-back switched off -- is rendered whole in
	 * the LAST chunk's launch (fast_voice: range_last), not chunk by chunk: found by round 5's drop-in sweep, 1 program of 3000) */
	if (fi.total == 0 || fi.bail || fi.total < vd.run_len || fi.seq == 1) { atomicOr(&P.work_count[1], 1u); return; }
	const uint32_t *ids = P.op_ids + vd.ops_ofs;
	VoiceOut vo;
	vo.pan_const = P.ops[ids[vd.carr_local]].line[L_PAN].v0; /* (a held line: finalize_kernel's advance leaves v0 alone) */
	vo.has_pan = vd.pan_dynamic_row != ~0u ? 1u : 0u;
	vo.valid_len = fi.total;
	vo.pan_row = vd.pan_dynamic_row;
	P.vinfo[vd.out_row] = vo;
}

/* Streams of a few voices whose last row is a look-back voice's and whose other rows are closed-form voices' (FastParams.tail_ok):
 * which of them the look-back launch mixes itself. A thread per stream, after analyze_kernel and ahead of both launches: every
 * row on the time-parallel path to the stream's end with a constant pan, the last one by the look-back build through its
 * oscillator step, the others by the closed-form launch that runs first. What happens later -- a hold that sends a voice to the
 * block loop, a repaired group -- shows in the mixer's guards, and mix_few_kernel then mixes the stream again (k_fast_group.h). */
__global__ void __launch_bounds__(64) tailmix_kernel(FastParams P, uint32_t n_streams) {
	const uint32_t s = blockIdx.x * 64 + threadIdx.x;
	if (s >= n_streams) return;
	const MixStream ms = P.inmix_stream[s];
	bool ok = ms.n_rows >= 1 && ms.n_rows <= 8 && ms.write_len > 0;
	const uint32_t last = ms.first_row + ms.n_rows - 1;
	for (uint32_t r = ms.first_row; ok && r <= last; ++r) {
		const FastInfo fi = P.info[r];
		const VoiceDesc vd = P.voices[r];
		if (fi.total != ms.write_len || fi.total != vd.run_len || fi.cub || fi.bail || vd.pan_dynamic_row != ~0u || vd.out_row != r) ok = false;
		if (fi.seq != (r == last ? 3u : 0u)) ok = false;
		if (vd.plan_len == 0 || P.steps[vd.plan_ofs + vd.plan_len - 1].kind == ST_VOICE) ok = false; /* (its carrier's step stores the row) */
	}
	P.tail_ok[s] = ok ? 1u : 0u;
	if (!ok) return;
	for (uint32_t r = ms.first_row; r <= last; ++r) { /* the rows' mixer records, as finalize_kernel will write them */
		const VoiceDesc vd = P.voices[r];
		VoiceOut vo;
		vo.pan_const = P.ops[(P.op_ids + vd.ops_ofs)[vd.carr_local]].line[L_PAN].v0; /* (a held line: finalize_kernel's advance leaves v0 alone) */
		vo.has_pan = 0u; vo.valid_len = P.info[r].total; vo.pan_row = ~0u;
		P.vinfo[r] = vo;
	}
	P.info[last].tail = ms.n_rows;
	P.info[last].tail_stream = s;
}

struct MixParams {
	const MixStream *streams;
	const float *vout;
	const float *pan;
	const VoiceOut *vinfo;
	uint32_t row_stride;
	uint32_t pcm_offset;
	uint32_t stereo;
	uint32_t swap_bytes; /* big-endian PCM for AU files (player/sndfile.c:160-168) */
	/* frame windows (premix_kernel above): an early launch mixes the 256-frame blocks [blk_lo, blk_hi); the last one skips
	 * the blocks below early_blocks when guard[0] (voices on the block loop's list) and guard[1] (premix_kernel's note) are 0 */
	uint32_t blk_lo, blk_hi, early_blocks;
	const uint32_t *guard;
	const uint32_t *inmix; /* the closed-form launch has mixed tiles itself (k_fast_types.h): its control words, or NULL */
	const uint32_t *tail_ok; /* [stream]: the look-back launch has mixed the stream itself (FastParams.tail_ok), or NULL */
};

/* generator.c:749-825: ordered voice sum (ref-build association) and PCM.
 * One thread per output frame walks the stream's voices in ascending id --
 * the reference's f32 accumulation order -- so the sum is bit-identical to
 * the CPU's and independent of scheduling.  Loads are issued eight voices
 * ahead of the (serially dependent) adds. */
#ifndef FK_TEMPORAL_ROWS /* the voice rows are read once: streaming loads (see FK_VSTORE in k_fast_voice.h) */
#define MIX_LOAD(p) __builtin_nontemporal_load(p)
#else
#define MIX_LOAD(p) (*(p))
#endif
constexpr int MIX_TILE = 256; /* voices whose constants are staged at a time */
constexpr int MIX_AHEAD = 32; /* loads per batch and thread */
template <int BLK = 256>
__device__ __forceinline__ void mix_body(const MixParams &P, const MixStream &ms, const uint32_t bx,
		float *s_pan, uint32_t *s_valid, uint32_t *s_prow, uint32_t &s_special, const bool covered = false) {
	const uint32_t tl = threadIdx.x; /* within the BLK frames of bx */
	const uint32_t i = bx * BLK + tl;
	const bool act = i < ms.write_len && !covered;
	float L = 0.f, R = 0.f;
	for (uint32_t r0 = 0; r0 < ms.n_rows; r0 += MIX_TILE) {
		const uint32_t nt = min((uint32_t)MIX_TILE, ms.n_rows - r0);
		__syncthreads();
		if (tl == 0) s_special = 0;
		__syncthreads();
		for (uint32_t q = tl; q < nt; q += BLK) {
			const VoiceOut vo = P.vinfo[ms.first_row + r0 + q];
			s_pan[q] = vo.pan_const;
			s_valid[q] = vo.valid_len;
			s_prow[q] = vo.has_pan ? vo.pan_row : ~0u;
			if (vo.has_pan || vo.valid_len < ms.write_len) s_special = 1;
		}
		__syncthreads();
		if (!act) continue;
		const float *base = P.vout + (size_t)(ms.first_row + r0) * P.row_stride + i;
		if (s_special == 0) {
			/* every row of the tile covers the whole segment with a constant pan */
			/* two batches of row loads in flight: the next batch is issued before the adds of this one (which are one
			 * dependent chain per channel, the reference's voice order) -- a segment of 44100 frames is only 689 waves,
			 * and what they keep in flight is what the HBM pipe sees */
			uint32_t r = 0;
			auto load = [&](float *sv, uint32_t at) {
#pragma unroll
				for (int u = 0; u < MIX_AHEAD; ++u) sv[u] = MIX_LOAD(&base[(size_t)(at + u) * P.row_stride]);
			};
			auto add = [&](const float *sv, uint32_t at) {
#pragma unroll
				for (int u = 0; u < MIX_AHEAD; ++u) {
					const float v = sv[u] * ms.amp_scale;
					const float s_r = v * s_pan[at + u];
					L = (L + v) - s_r;
					R = (R + v) + s_r;
				}
			};
			float sa[MIX_AHEAD], sb[MIX_AHEAD];
			if (r + MIX_AHEAD <= nt) load(sa, r);
			while (r + MIX_AHEAD <= nt) {
				const bool more_b = r + 2 * MIX_AHEAD <= nt;
				if (more_b) load(sb, r + MIX_AHEAD);
				add(sa, r);
				r += MIX_AHEAD;
				if (!more_b) break;
				if (r + 2 * MIX_AHEAD <= nt) load(sa, r + MIX_AHEAD);
				add(sb, r);
				r += MIX_AHEAD;
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

template <int BLK>
__device__ __forceinline__ void mix_kernel_body(const MixParams &P) {
	__shared__ float s_pan[MIX_TILE];
	__shared__ uint32_t s_valid[MIX_TILE];
	__shared__ uint32_t s_prow[MIX_TILE]; /* pan row, or ~0u */
	__shared__ uint32_t s_special;        /* tile has a short row or a pan row */
	const MixStream ms = P.streams[blockIdx.y];
	const uint32_t bx = blockIdx.x + P.blk_lo; /* (blk_lo, blk_hi, early_blocks: in blocks of 256 frames -- the 256-frame form's only) */
	if (bx * BLK >= ms.write_len || (P.blk_hi && bx >= P.blk_hi)) return;
	if (P.blk_hi && P.guard[1]) return; /* an early launch, and premix_kernel could not speak for every voice: their records are not there yet */
	if (P.early_blocks && bx < P.early_blocks && P.guard[0] == 0 && P.guard[1] == 0) return; /* mixed by an early launch, and it stands */
	bool covered = false;
	if (P.inmix && P.guard[0] == 0 && P.guard[1] == 0) {
		/* frames the closed-form launch has mixed itself: those of the tiles with their bits set (and it stands) */
		const uint32_t cf = P.inmix[INMIX_CF], i = bx * BLK + threadIdx.x;
		const uint32_t base = P.inmix[INMIX_BASE], cfs = P.inmix[INMIX_CFS]; /* (the short chunks at the end: k_fast_types.h) */
		uint32_t k = 0, j = 0;
		if (i < base || !cfs) { k = cf ? i / cf : 0u; j = cf ? (i - k * cf) / INMIX_TILE : 0u; }
		else { const uint32_t ks = (i - base) / cfs; k = P.inmix[INMIX_NCH1] + ks; j = (i - base - ks * cfs) / INMIX_TILE; }
		covered = k < P.inmix[INMIX_NCH] && ((P.inmix[INMIX_CHUNK + INMIX_LINE * k + INMIX_BITS + (j >> 5)] >> (j & 31u)) & 1u) != 0;
		if (__syncthreads_and(covered)) return;
	}
	mix_body<BLK>(P, ms, bx, s_pan, s_valid, s_prow, s_special, covered);
}
__global__ void __launch_bounds__(256) mix_kernel(MixParams P) { mix_kernel_body<256>(P); }
/* ... in blocks of 64 frames, a wave each: what follows a closed-form launch that has mixed most of the tiles itself (round 6). What
 * is left then is a few hundred tiles in a row -- each XCD's last chunk -- and as workgroups of 256 frames x 1024 rows (1 MB of
 * loads each) they were one or two per CU, by chance: 87 us for 0.36 GB. A quarter the size they spread evenly */
__global__ void __launch_bounds__(64) mix_kernel64(MixParams P) { mix_kernel_body<64>(P); }

/* Streams of a few voices each (a batch of many small scripts: BASELINE config 4 has two voices per render): four
 * consecutive frames per thread, 16-byte row loads, per-row records read straight from memory -- mix_kernel's tile
 * staging (three barriers per workgroup for a tile of two rows) and one workgroup per 256 frames left the 64-stream
 * batch's mixer at 1 TB/s. Same sums in the same order. pcm_offset is a multiple of 4 (the host checks). */
__global__ void __launch_bounds__(256) mix_few_kernel(MixParams P) {
	const MixStream ms = P.streams[blockIdx.y];
	const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
	if (i0 >= ms.write_len) return;
	/* mixed by the launch that rendered its last row, and nothing has touched a row since (no voice on the block loop's list, no
	 * repaired group): done */
	if (P.tail_ok && P.tail_ok[blockIdx.y] && P.guard[0] == 0 && P.guard[1] == 0) return;
	float L[4] = {0.f, 0.f, 0.f, 0.f}, R[4] = {0.f, 0.f, 0.f, 0.f};
	const bool full = i0 + 4 <= ms.write_len;
	for (uint32_t r = 0; r < ms.n_rows; ++r) {
		const VoiceOut vo = P.vinfo[ms.first_row + r];
		const float *row = P.vout + (size_t)(ms.first_row + r) * P.row_stride + i0;
		float v4[4], p4[4] = {vo.pan_const, vo.pan_const, vo.pan_const, vo.pan_const};
		bool ok[4];
		if (full && i0 + 4 <= vo.valid_len) {
			typedef float __attribute__((ext_vector_type(4))) f32x4;
#ifndef FK_TEMPORAL_ROWS
			const f32x4 q = __builtin_nontemporal_load((const f32x4 *)row);
#else
			const f32x4 q = *(const f32x4 *)row;
#endif
			v4[0] = q.x; v4[1] = q.y; v4[2] = q.z; v4[3] = q.w;
			ok[0] = ok[1] = ok[2] = ok[3] = true;
		} else {
#pragma unroll
			for (int k = 0; k < 4; ++k) { ok[k] = i0 + k < ms.write_len && i0 + k < vo.valid_len; v4[k] = ok[k] ? row[k] : 0.f; }
		}
		if (vo.has_pan) {
			const float *prow = P.pan + (size_t)vo.pan_row * P.row_stride + i0;
#pragma unroll
			for (int k = 0; k < 4; ++k) if (ok[k]) p4[k] = prow[k];
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if (ok[k]) { /* generator.c:842-843: a voice adds only the frames it produced */
				const float v = v4[k] * ms.amp_scale;
				const float s_r = v * p4[k];
				L[k] = (L[k] + v) - s_r;
				R[k] = (R[k] + v) + s_r;
			}
		}
	}
	int16_t o[8];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		if (P.stereo) {
			const int16_t l16 = pcm16(L[k]), r16 = pcm16(R[k]);
			o[2 * k] = P.swap_bytes ? pcm_swap(l16) : l16;
			o[2 * k + 1] = P.swap_bytes ? pcm_swap(r16) : r16;
		} else {
			const int16_t m16 = pcm16((L[k] + R[k]) * 0.5f);
			o[k] = P.swap_bytes ? pcm_swap(m16) : m16;
		}
	}
	if (P.stereo) {
		int16_t *d = ms.pcm + 2 * (size_t)(P.pcm_offset + i0);
		if (full) *(uint4 *)d = *(const uint4 *)o;
		else for (uint32_t k = 0; i0 + k < ms.write_len; ++k) { d[2 * k] = o[2 * k]; d[2 * k + 1] = o[2 * k + 1]; }
	} else {
		int16_t *d = ms.pcm + (size_t)(P.pcm_offset + i0);
		if (full) *(uint2 *)d = *(const uint2 *)o;
		else for (uint32_t k = 0; i0 + k < ms.write_len; ++k) d[k] = o[k];
	}
}

/* (launched with 64 threads per workgroup: told so, the kernel keeps a DevOp in registers instead of spilling 84 of them) */
__global__ void __launch_bounds__(64) event_kernel(DevOp *ops, const OpUpdate *recs, uint32_t n, const WaveConst *wc) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	OpUpdate u = recs[i];
	DevOp o = ops[u.op];
	apply_update(o, u, wc);
	ops[u.op] = o;
}
