/* k_fast_types.h -- part of hip_backend.hip (included there, inside namespace sauhip; not a header of its own).
 * The time-parallel path's parameter block, per-voice analysis record and constants. */
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
	uint32_t early;   /* something that runs before chain_kernel's chunks -- a running sum, another chain's inputs -- depends on
	                   * the output of a chain of this voice, and every chain it depends on is fed from its own lines
	                   * (step_is_chain_inline): those chains run first, whole segment, in a chain_kernel launch ahead of the
	                   * sum passes (FastParams.chain_early), and every pass reads their samples from the rows (FT_CHAIN_EARLY) */
	uint32_t cub;     /* a closed-form voice with an R oscillator of `cub` segments and the reference's loop tails on: rendered by
	                   * the closed-form build with the tail code, fast_kernel<4, 0, true>, and by no other launch */
	uint32_t tail;    /* round 6 (tailmix_kernel, k_finish.h): this voice is the last row of a stream of a few voices whose other rows the
	                   * launch before has written, and the look-back launch that renders it mixes the stream as it stores: the rows
	                   * of the stream (this one included), or 0 */
	uint32_t tail_stream; /* ... and the stream's index (FastParams.inmix_stream[]) */
};

/* A wave table in the time-parallel kernels' LDS: per table one block -- [c3, c2] x 2048 (f64 pairs), then [c1, c0] x 2048.
 * Two forms (a template parameter of the kernels, WIDE). Narrow: [c1, c0] as the f32 pairs they are in HBM (sau_dev_math.h,
 * HerpC01), 48 KiB per table, two conversions per sample. Wide (round 4): widened to f64 when the block is staged, 64 KiB per
 * table -- no conversion per sample, and with a power-of-two block one address computation serves both reads: 3 of the 38
 * VALU instructions per operator-sample of config 3 for 8 more bytes of LDS gather per sample. It pays where the launch
 * keeps its rows per pass (config 3: 2.03 -> 1.97 ms per launch) and costs where the wider blocks push rows out of LDS
 * (carrier-FM bank 6 -> 5 rows: +3.7 %, config 4 +9 %; profiles/r04_headline_ab.json) -- so only the closed-form builds at 8
 * and 6 rows have a wide form, and the host picks it only when every table the segment wants still fits at those rows. */
template <bool WIDE> struct FkTab {
	static constexpr uint32_t BYTES = WIDE ? 65536u : (uint32_t)(WAVE_LEN * (sizeof(HerpC23) + sizeof(HerpC01)));
	static constexpr uint32_t C01 = (uint32_t)(WAVE_LEN * sizeof(HerpC23)); /* where a block's [c1, c0] entries begin */
};
constexpr uint32_t FAST_TAB_BYTES = FkTab<false>::BYTES, FAST_TAB_BYTES_WIDE = FkTab<true>::BYTES;

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
constexpr uint32_t FAST_FLAGS = FAST_MAX_LEVELS + 9; /* pass_flags words */
constexpr uint32_t FAST_EARLY_FLAG = FAST_MAX_LEVELS + 8; /* some voice has early chains (FastInfo.early) */
constexpr uint32_t FAST_CUB_FLAG = FAST_MAX_LEVELS + 7; /* some voice has FastInfo.cub set */
constexpr uint32_t FAST_CUB_ROWS = 4;                  /* rows per pass of that build */
constexpr uint32_t FAST_CF_COUNT = FAST_MAX_LEVELS + 5; /* voices in FastParams.vlists[0]: closed-form ones of a segment that also has look-back voices */
constexpr uint32_t FAST_LK_COUNT = FAST_MAX_LEVELS + 6; /* ... in vlists[1]: the look-back voices */
constexpr uint32_t FAST_LEAN_FLAG = FAST_MAX_LEVELS + 4; /* ... some voice has feedback chains and no running sum to scan (fast_kernel<T, 3>) */
constexpr uint32_t FAST_DYN_CTR = FAST_MAX_LEVELS + 3; /* ... the one that deals out fast_kernel<T, 0>'s tasks (dyn_chunks) */
/* Decoded steps are kept once per pass that runs them ([list][voice][step]): a pass walks its own list and never
 * loads a step only to find that another pass needs it (the per-step cost of the interpreter is most of a pass). */
constexpr uint32_t FAST_LISTS = 5; /* 0: only / final pass, 1..3: sum passes, 4: chain-input pass */
__device__ __forceinline__ uint32_t fast_list_of(uint32_t mode, uint32_t sum_levels) {
	return (mode == 0 || mode == sum_levels + 1) ? 0u : (mode == sum_levels + 2 ? 4u : mode);
}
constexpr uint32_t FR_CHAIN_IN = 4u << FAST_MAX_LEVELS; /* FastStep.ramp: the chain-input pass runs this step */
constexpr uint32_t FR_FINAL_SKIP = 8u << FAST_MAX_LEVELS; /* ... the final pass does not: only chains' inputs needed it */
constexpr uint32_t FT_CHAIN = 1u << 18;     /* FastStep.type: a feedback chain (rows = bits of FastStep.pan) */
constexpr uint32_t FT_CHAIN_EARLY = 1u << 20; /* ... whose samples are in its row before any pass runs (FastInfo.early) */
constexpr uint32_t FT_CUBTAIL = 1u << 19;   /* ... an R oscillator with `cub` segments and the reference's loop tails on: FastStep.phase0 =
                                             * frames until it, an ancestor or the voice stops (TailCtx.rem) */
constexpr uint32_t CHAIN_MARK = 0xC4A10001u; /* DevOp.ras_level of a W operator: chain_kernel staged its state */

/* Chunk c's run of row groups [lo, hi) of a voice's ngroups: K chunks asked for by the host; small > 0: the last INMIX_NSMALL of
 * them `small` groups long, the others sharing what is in front (and numbered first). Used alike by fast_voice (a task's groups),
 * premix_kernel (the chunks' frames) and, through its control words, the tiles and mix_kernel. -> regular chunks there are, the
 * group the short ones begin at (ngroups: none), groups per regular chunk */
constexpr uint32_t INMIX_NSMALL = 8; /* short chunks at the end of the closed-form launch's queues: one per XCD (see INMIX_NCH1 below) */
struct FkChunks { uint32_t nch1, body, per; };
__device__ __forceinline__ FkChunks fk_chunk_groups(const uint32_t ngroups, const uint32_t K, const uint32_t small, const uint32_t c,
		uint32_t &lo, uint32_t &hi) {
	FkChunks q;
	if (!small || K <= INMIX_NSMALL || ngroups <= 2 * INMIX_NSMALL * small) {
		q.per = K ? (ngroups + K - 1) / K : ngroups;
		q.body = ngroups;
		q.nch1 = q.per ? (ngroups + q.per - 1) / q.per : 0u;
		lo = c * q.per;
		if (lo > ngroups) lo = ngroups;
		hi = lo + q.per < ngroups ? lo + q.per : ngroups;
		return q;
	}
	q.body = ngroups - INMIX_NSMALL * small;
	q.per = (q.body + (K - INMIX_NSMALL) - 1) / (K - INMIX_NSMALL);
	q.nch1 = (q.body + q.per - 1) / q.per;
	if (c < q.nch1) {
		lo = c * q.per;
		hi = lo + q.per < q.body ? lo + q.per : q.body;
	} else {
		lo = q.body + (c - q.nch1) * small;
		if (lo > ngroups) lo = ngroups;
		hi = lo + small < ngroups ? lo + small : ngroups;
	}
	return q;
}

struct MixStream {
	uint32_t first_row, n_rows;
	float amp_scale;
	uint32_t write_len;
	int16_t *pcm; /* stream's PCM row */
};
/* The mixer inside the closed-form launch (round 5). A bank of many voices mixed into one stream: the mixer reads every voice
 * row once more -- 1.8 GB, 0.26 ms at 6.9 TB/s, an eighth of a BASELINE config-3 step -- while the launch that wrote them is
 * bound by vector issue and leaves HBM idle. Mixed by the launch itself, the reads hide under its arithmetic -- but the six
 * flops per voice-sample that the mixer hides under its HBM time are then vector instructions of an issue-bound launch
 * (+3 %), so the gain is what is left of the mixer's 0.2 ms: 0 to 3 % of a config-3 step by box. Voice order
 * is the reference's f32 sum order (generator.c:749-825) and cannot be split, so a tile of INMIX_TILE frames is mixed by one
 * wave over all rows, as mix_kernel does it, once every voice has written those frames:
 *  - the launch's tasks (voice, run of row groups = chunk) are dealt out chunk-major instead of voice-major, one queue per
 *    XCD: chunk k belongs to XCD k mod 8, whose waves render it for every voice and mix its tiles. Producer and consumer
 *    share an L2, so the hand-off needs no L2 write-back (MI355X_MICROARCH.md, inter-workgroup visibility): the storing
 *    wave waits for its stores (vmcnt) and adds to the chunk's counter; the mixing wave reads the counter, invalidates its
 *    L1 (agent-scope acquire) and loads;
 *  - who mixes what is fixed, not claimed: some of a chunk's tasks each mix, when their voice is rendered, a tile of the XCD's
 *    chunk before (eight chunks back), which the queue has dealt out whole by then and a task's length ago; if its counter
 *    is not full after all, the tile is left alone. A finished tile sets its bit. (Tiles taken from a shared counter, first version:
 *    three dependent round trips per task to find one -- the launch's waves stall for them, and four waves per SIMD
 *    hide nothing: 1.80 -> 1.95 ms per launch for 0.18 ms less mixing.)
 *  - a wave whose XCD has run out of tasks takes another XCD's (nobody idles at the end); such a task counts for nothing
 *    and mixes nothing, so a chunk it touched stays incomplete;
 *  - mix_kernel runs after the launch as before and mixes every frame whose tile has no bit -- each XCD's last chunk at
 *    least; all of them when premix_kernel or the pass itself says the early results do not stand (k_finish.h).
 * premix_kernel decides (work_count[1]): every voice on this path to its end, same lead-in (so the chunks' frame ranges
 * are the same for all), constant pan, no shorter than the stream. Control words (FastParams.inmix): */
constexpr uint32_t INMIX_TILE = 256;        /* frames per tile: four per lane, 16-byte row loads */
/* (words that many waves add to or poll lie on 128-byte lines of their own: an agent-scope atomic takes its line for about
 * 12 ns chip-wide, and with the queues' counters and the constants all on one line the launch took 5.2 ms for 2.05) */
constexpr uint32_t INMIX_LINE = 32;         /* words per line */
constexpr uint32_t INMIX_CF = 0;            /* [0]: frames per chunk, [1]: tiles per chunk, [2]: chunks (premix_kernel; read-only in the launch) */
constexpr uint32_t INMIX_TPC = 1, INMIX_NCH = 2;
/* Round 6: the queues' last chunks are short ones. A chunk's tiles are mixed by the tasks of the XCD's next chunk, so each XCD's
 * last chunk is left to mix_kernel -- 8 of config 3's 48 chunks, 65 us of a 2 ms step. With the last eight chunks a quarter as
 * long (FastParams.dyn_small row groups each) it is 8 short ones of 56. [3]: chunks of the regular length (they come first),
 * [4]: the frame the short ones begin at (~0u: there are none), [5]: frames per short chunk */
constexpr uint32_t INMIX_NCH1 = 3, INMIX_BASE = 4, INMIX_CFS = 5;
constexpr uint32_t INMIX_QUEUE = 1 * INMIX_LINE; /* + INMIX_LINE x: XCD x's task counter */
constexpr uint32_t INMIX_CHUNK = 9 * INMIX_LINE; /* + INMIX_LINE k: chunk k's line -- [0] own-XCD tasks that have stored their rows, [8 + j / 32] bit j % 32: tile j is mixed */
constexpr uint32_t INMIX_DONE = 0, INMIX_BITS = 8;
constexpr uint32_t INMIX_MAX_TPC = 256;     /* tiles per chunk at most (eight words of bits) */
constexpr uint32_t INMIX_MAX_CHUNKS = 2039;
constexpr uint32_t INMIX_WORDS = INMIX_CHUNK + INMIX_LINE * INMIX_MAX_CHUNKS; /* 256 KiB */
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
	uint32_t n_chain_slots; /* lanes of the chain kernels (entries of chain_desc): the rows, or more where kinds of R feedback begin waves of their own */
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
	uint32_t chain_early_ok; /* analyze_kernel may mark voices early (SAU_AMD_NO_EARLY_CHAINS=1: off -- such voices go to the block loop) */
	uint32_t chain_early; /* chain_kernel: 1: this launch runs the early chains (CL_EARLY), whole segment; 0: the others */
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
	uint32_t look_wpv_flags; /* 1: no rings in LDS (tuning aid) */
	uint32_t look_wpv;  /* waves per voice in fast_kernel<T, 2>'s launches (1..64): a voice whose waves sit in one workgroup
	                     * looks back through rings in LDS, one spread over neighbouring workgroups through words in HBM */
	uint32_t rows_multi; /* rows per pass in the launches of the full running-sum build (kinds 1 and 2) */
	/* The closed-form build's launch deals its work out through a counter (pass_flags[FAST_DYN_CTR], zero at the start of
	 * every segment: finalize_kernel): task = (voice, one of dyn_chunks runs of consecutive row groups). Waves take the
	 * next task when they finish one, so CUs that get less done (other kernels' workgroups sharing them: the previous
	 * segment's mixer) hold nobody up at the end. 0: static shares (waves stride over voices and groups). */
	uint32_t dyn_chunks;
	uint32_t dyn_static; /* 1: the same tasks in fixed strides over the launch's waves, no counter (SAU_AMD_NO_DYN) */
	uint32_t dyn_small;  /* > 0: the last eight of the dyn_chunks runs are this many row groups long (fk_chunk_groups) */
	/* Voices with feedback chains whose other oscillators all have closed-form phases (a chain that sums its own
	 * increments counts as such: BASELINE config 5) take no sum pass, no scan and no saved increments: their
	 * chain-input and final passes run in a build of their own, fast_kernel<T, 3> -- fast_voice without the code of
	 * the several-pass sums, rows_lean rows per pass. lean_on: such launches exist (the full build leaves those voices out). */
	uint32_t rows_lean, lean_on;
	/* A segment whose voices may have running sums (the host cannot tell: analyze_kernel decides per voice) is rendered
	 * by two launches since round 3, each over a list of voices analyze_kernel builds: the closed-form voices by the
	 * closed-form build -- rows_cf rows per pass, block buffers without frequency blocks, tasks dealt out by the counter
	 * -- and the look-back voices by the single-pass build, which then chooses the waves per voice from how many such
	 * voices there are (look_words_real: the words in HBM exist, so a voice may spread over workgroups). Together in
	 * one launch they had the same number of waves per voice whatever a voice cost: BASELINE config 4's two voices
	 * per render (one closed-form, one with nested running sums) took 6.0 ms where they take 1.6 + 3.1 ms apart. */
	uint32_t *vlists;     /* [2][n_voices], or NULL (one launch over every voice) */
	uint32_t split_cf, rows_cf, look_words_real, look_groups;
	uint32_t cub_ok;      /* the build with the `cub` tails will be launched (it fits LDS): FastInfo.cub voices may stay on this path */
	uint32_t only_multi; /* this launch: only the voices fast_kernel<T, 2> leaves out (one wave in order, several passes) */
	int8_t ctab_of_wave[12];
	uint8_t cwave_of_tab[12];
	int8_t tab_of_wave[12];
	uint8_t wave_of_tab[12];
	WaveConst wc[12];
	/* (at the end: the 12-row closed-form build fits its 128 vector registers exactly, and with these 24 bytes in front of the
	 * tables above -- every offset behind them moved -- it spilled 88 of them, with not a line of the mixing code compiled in) */
	uint32_t *inmix;      /* the mixer inside the closed-form launch: INMIX_WORDS control words, or NULL */
	const MixStream *inmix_stream; /* ... the one stream */
	uint32_t inmix_div_m, inmix_div_s; /* the voice count as a divisor: q / n_voices = udiv_magic(q, m, s), scalar instructions only (a task's number -> chunk, voice) */
	uint32_t inmix_flags, inmix_pcm_offset; /* 64: this launch takes its tasks from the XCDs' queues; 32: ... and mixes (premix_kernel has run); 1: stereo PCM,
	                                         * 2: byte-swapped; 4: timing aid; bits 8-11: which tasks mix (sixteenths into a chunk). MixParams.pcm_offset */
	/* Streams of a few voices each (a batch of small scripts: BASELINE config 4 has a closed-form and a look-back voice per render):
	 * the look-back launch mixes such a stream while it stores the stream's last row (FastInfo.tail) -- per frame it reads the
	 * rows before from HBM (written by the closed-form launch that ran first), adds its own sample in the mixer's order and writes
	 * the PCM; mix_few_kernel then skips the stream unless the guards say a row changed afterwards. tail_ok: one word per stream
	 * (tailmix_kernel's verdict), or NULL; inmix_stream is then the streams' array; tail_flags: 1 stereo PCM, 2 byte-swapped */
	uint32_t *tail_ok;
	uint32_t tail_flags, tail_pcm_offset;
	uint32_t edge_only; /* the 12-row wide closed-form build: nonzero = the first and the last row group of every voice only (the launch ahead
	                     * of fast_kernel<12, 0, false, true, false, true>, which renders the groups between) */
};
