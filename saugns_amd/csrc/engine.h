/* engine.h -- host control plane of the MI355X generator backend.
 *
 * Mirrors the control flow of sau/generator.c (event timeline, voice
 * durations, end-of-signal detection: generator.c:172-195,348-377,833-973)
 * on the host, flattens every voice's operator graph into a plan
 * (plan.cpp), and drives a Backend that owns the operator state and does the
 * per-sample work.  The product backend is the HIP one (hip_backend.hip);
 * the interface exists so that the host logic can be exercised without a GPU
 * by tests/ (which inject their own sequential executor) -- the library
 * itself never falls back to a CPU path.
 */
#ifndef SAU_ENGINE_H
#define SAU_ENGINE_H

#include "../../include/sau_abi.h"
#include "sau_dev_types.h"
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <unordered_map>
#include <vector>

namespace sauengine {

using namespace saudev;

/* Everything the backend needs to render one segment (no events inside). */
struct SegmentDesc {
	uint32_t len;             /* frames */
	uint32_t pcm_offset;      /* frame offset into each stream's PCM row */
	bool stereo;
	bool swap_bytes;          /* store PCM big-endian (AU files, player/sndfile.c:160-168) */
	/* active voices of all streams, ascending (stream, voice id) */
	const VoiceDesc *voices;
	uint32_t n_voices;
	/* per stream: range of `voices`, amp_scale, frames to write (last_len) */
	struct Stream {
		uint32_t first_voice, n_voices;
		float amp_scale;
		uint32_t write_len;
	};
	const Stream *streams;
	uint32_t n_streams;
	uint32_t n_slots;         /* max over active plans of main + frequency slots */
	uint32_t n_main;          /* max main-pool slots (slot_index() base) */
	uint32_t n_fast;          /* max block buffers the time-parallel path needs (fast_slot_compact) */
	uint32_t n_fast_full;     /* the same with frequency blocks, over voices that may need them */
	bool may_scan;            /* some voice may have ramped or modulated frequencies (running-sum phases) */
	uint32_t sum_levels = 2; /* running sums nested that deep may occur (host estimate; the device checks) */
	uint32_t max_ops;         /* max operators in any active voice */
	uint32_t max_steps;       /* longest active plan */
	uint32_t n_pan_rows;
	uint64_t wave_mask;       /* wave ids in use (bit per id) */
	bool maybe_block;         /* some voice may need the block loop (sweeps, FM, ...) */
	bool serial;              /* some voice may run a per-sample feedback recurrence (self-modulation) */
	bool maybe_cub = false;   /* some voice has an R operator with `cub` segments and the loop tails are on (the builds with that code are launched) */
	uint32_t n_chain_rows = 0;/* row pairs the voices' chain_base/n_chain span */
	uint32_t n_inc_rows = 0;  /* row pairs the voices' inc_base/n_inc span (voices that may have running-sum phases) */
	uint32_t n_look_rows = 0; /* rows the voices' look_base/n_look span (those of them without feedback chains) */
	uint32_t n_may_scan = 0;  /* voices that may have running-sum phases this segment (the device decides: analyze_kernel) */
	uint32_t n_chain_slots = 0; /* lanes of the chain kernels' waves the voices' chain_slot/n_chain span (>= n_chain_rows) */
	bool chain_rows_padded = false; /* some of the n_chain_slots lanes belong to no voice (kinds of R feedback chains begin waves of their own) */
};

/* Bytes a backend has for the rows of a segment's feedback chains (8 B per chain and frame; the engine cuts segments with
 * feedback voices accordingly, never below CHAIN_SEG frames). free_bytes: what was free on the backend's device when the
 * process first opened it (0: not told -- the sequential executor of the tests); alloc_failures: allocations of rows that
 * failed on this backend. SAU_AMD_CHAIN_ROWS_MB sets the budget; without it: 24 GiB (288 GB of HBM hold BASELINE config 5's
 * 4096 chains x 441000 frames, 14.4 GB, as one segment -- one fill and one drain of the pass / chain pipeline),
 * or an eighth of free_bytes where that is less; every failed allocation halves it (the backend renders the segment that did
 * not fit in the block loop). Per backend since round 6 (Backend::chain_rows_budget): the hint used to be the first device's
 * of the process and the failures every engine's -- one full GPU of eight cut every other GPU's segments short (ADVICE r04,
 * VERDICT r05). (inline: the test executor's library has the interface without engine.cpp) */
inline size_t chain_rows_budget(size_t free_b, unsigned fails) {
	static const long long env_mb = [] {
		const char *v = getenv("SAU_AMD_CHAIN_ROWS_MB");
		return v ? atoll(v) : -1ll;
	}();
	const unsigned sh = fails > 40 ? 40 : fails; /* (after a few failures: segments with feedback voices are CHAIN_SEG frames) */
	if (env_mb >= 0) return ((size_t)env_mb << 20) >> sh;
	const size_t dflt = (size_t)24 << 30; /* of an MI355X's 288 GB */
	if (!free_b) return dflt >> sh;
	return (free_b / 8 < dflt ? free_b / 8 : dflt) >> sh;
}

struct BackendConfig {
	uint32_t srate;
	uint32_t op_count;        /* all streams */
	uint32_t voice_count;     /* all streams */
	uint32_t n_streams;
	uint32_t max_frames;      /* PCM row capacity, frames */
	const float *piluts;      /* 12 x 2048 */
	const WaveConst *wconst;  /* 12 */
};


class Backend {
public:
	virtual ~Backend() {}
	virtual bool init(const BackendConfig &cfg, std::string &err) = 0;
	/* grow PCM / voice matrices when a caller passes a longer buffer */
	virtual bool reserve_frames(uint32_t max_frames, bool stereo, std::string &err) = 0;
	/* plans changed: full step array + op id lists; fast_ids holds 2 * n_steps entries:
	 * the numbering without frequency blocks, then the one with them */
	virtual bool upload_plans(const Step *steps, const FastIds *fast_ids, size_t n_steps,
			const uint32_t *op_ids, size_t n_ids, std::string &err) = 0;
	/* apply operator updates in order; ops are distinct within one call */
	virtual bool apply_updates(const OpUpdate *recs, size_t n, std::string &err) = 0;
	/* zero the frames [first_frame, first_frame + n_frames) of the PCM rows of streams [first_stream, first_stream + n_streams):
	 * what a run's mixers do not write (generator.c:911-914 clears the whole buffer ahead of every call; here only the frames
	 * behind a stream's last voice and the segments nothing sounds in are cleared, ordered with the rendering like any device work) */
	virtual bool zero_pcm(uint32_t first_stream, uint32_t n_streams, uint32_t first_frame, uint32_t n_frames, bool stereo,
			std::string &err) = 0;
	virtual bool render(const SegmentDesc &seg, std::string &err) = 0;
	/* copy stream `s` PCM [0, frames) to host memory (blocks until done) */
	virtual bool fetch_pcm(uint32_t stream, int16_t *dst, uint32_t frames,
			bool stereo, std::string &err) = 0;
	/* device address of stream s PCM row, or NULL (test backends) */
	virtual const int16_t *device_pcm(uint32_t stream) = 0;
	virtual bool sync(std::string &err) = 0;
	/* Output stage: queue a copy of stream s PCM [0, frames) into host memory from
	 * alloc_host(); `slot` (0..3) names the copy for wait_fetch(). Later device
	 * work is ordered behind the copy. Defaults suit CPU test backends. */
	virtual bool fetch_pcm_async(uint32_t stream, int16_t *dst, uint32_t frames, bool stereo,
			int slot, std::string &err) { (void)slot; return fetch_pcm(stream, dst, frames, stereo, err); }
	virtual bool wait_fetch(int slot, std::string &err) { (void)slot; (void)err; return true; }
	virtual void *alloc_host(size_t bytes) { return malloc(bytes); }
	virtual void free_host(void *p) { free(p); }
	/* Operator state kept / brought back (Engine::snapshot / restore: the drop-in generator's read-ahead rewinds when
	 * its host changes the size or the channel layout of its calls). `slot` (0..3) names the copy; both are ordered
	 * with the rendering like any other device work. */
	virtual bool save_state(int slot, std::string &err) { (void)slot; err = "this backend keeps no state snapshots"; return false; }
	virtual bool load_state(int slot, std::string &err) { (void)slot; err = "this backend keeps no state snapshots"; return false; }
	/* bytes this backend has for the feedback chains' rows of one segment (engine.h: chain_rows_budget; the engine cuts segments
	 * with such voices accordingly) */
	virtual size_t chain_rows_budget() { return sauengine::chain_rows_budget(0, 0); }
};

/* ---- plan compiler (plan.cpp) -------------------------------------------- */

struct OpMirror { /* host-side knowledge about one operator */
	bool inited = false;
	uint8_t type = 0;
	uint32_t time = 0;
	bool time_inf = false;
	uint8_t line_set = 0;             /* bit per L_*: line ever given */
	LineState pan;                    /* mirrored exactly (never ratio-scaled) */
	const sauProgramIDArr *mods[SAU_POP_NAMED] = {}; /* by use type; [0] unused */
	uint8_t wave = 0;
	bool ras_cub_seen = false;        /* an R operator that was ever given `cub` segments (conservative: SegmentDesc.maybe_cub) */
	uint32_t ras_kind = 0;            /* an R operator's line shape | function << 8 | flags << 16 as last given (the order feedback chains are dealt to waves in) */
	bool goal_seen = false;           /* some event gave one of its lines a sweep */
	bool freq_goal_seen = false;      /* ... one of its frequency lines */
	bool freq_ratio_seen = false;     /* some event gave one of its frequency lines a ratio (state or goal) */
	OpMirror() { pan = LineState{0, 0, 0, 0, 0, 0}; }
};

struct VoicePlan {
	std::vector<Step> steps;       /* (empty in a voice that uses its shape's: Engine::shapes_) */
	uint32_t n_steps = 0;          /* length of the step list the voice runs */
	std::vector<uint32_t> op_ids;  /* voice-local index -> stream-local op id */
	uint32_t carr_local = 0;
	uint32_t n_slots = 0;          /* memory slots: main pool + frequency pool */
	uint32_t n_main = 0;           /* main-pool slots (ids below FSLOT_BASE) */
	uint32_t n_fast = 0;           /* buffers live at once in the time-parallel path */
	std::vector<FastIds> fast_ids; /* per step: block buffers renumbered by liveness */
	uint32_t n_fast_full = 0;      /* ... counting frequency blocks too (ramps, FM) */
	std::vector<FastIds> fast_ids_full;
	uint64_t wave_mask = 0;
	bool has_camods = false;
	bool no_fast = false;          /* an operator is evaluated twice per block */
	bool wide = false;             /* step pairs with 16-bit buffer ids (sau_dev_types.h: wide plans); no_fast too */
	bool static_block = false;     /* graph has FM / feedback / R / filtered noise: block loop */
	bool selfmod = false;          /* a self-modulation amount has modulators of its own */
	bool ras_cub = false;          /* an R operator that was ever given `cub` segments */
	uint32_t n_chain = 0;          /* oscillator steps that may run a feedback recurrence (step_may_chain) */
	uint32_t n_osc = 0;            /* W and R oscillator steps (their phase increments may be saved between passes) */
};

/* Flatten the graph under `carrier` into steps. Returns false (with err) when
 * the graph exceeds what a workgroup can hold. */
bool compile_voice_plan(const std::vector<OpMirror> &ops, uint32_t carrier,
		VoicePlan &out, std::string &err);
/* the graph's shape as a token stream + its operators in plan order (plan.cpp); false: not cacheable */
bool voice_plan_shape(const std::vector<OpMirror> &ops, uint32_t carrier, std::vector<uint32_t> &tokens,
		std::vector<uint32_t> &op_ids, std::vector<uint32_t> &stamp, uint32_t mark, uint64_t &wave_mask);

/* ---- engine ---------------------------------------------------------------- */

/* Segments with feedback voices are at most this long: the recurrences' inputs and outputs pass
 * through per-chain rows in HBM, sized for one segment. */
constexpr uint32_t CHAIN_SEG = 131072;
/* Environment switches. Product settings are read as they are (INTEGRATION.md has the table: SAU_AMD_DEVICE,
 * SAU_AMD_READAHEAD*, SAU_AMD_LOOP_TAILS, SAU_AMD_CHAIN_ROWS_MB, SAU_AMD_POOL_MB, SAU_AMD_PINNED_POOL_MB, SAU_AMD_DEBUG*);
 * every other SAU_AMD_* name is a tuning or test aid and is looked at only when SAU_AMD_TUNE is set -- a stray variable
 * in a host's environment cannot change how, or on which kernels, a render runs. */
const char *tune_env(const char *name);
uint32_t chain_seg_frames(size_t n_chains, size_t budget_bytes);

class Engine {
public:
	/* programs are borrowed and must outlive the engine (generator.c:191) */
	static Engine *create(const sauProgram *const *prgs, size_t n_prgs,
			uint32_t srate, Backend *backend /* owned */, std::string &err);
	~Engine();

	/* Advance every stream by buf_len frames. host_bufs[s] may be NULL (PCM
	 * stays on the device). more[s]/out_len[s] as sauGenerator_run. */
	bool run(int16_t *const *host_bufs, size_t buf_len, bool stereo,
			bool *more, size_t *out_len, std::string &err);

	/* Size the device buffers for runs of up to `frames` frames now (run() grows them on demand, which waits for the
	 * stream: a host that knows its longest run says so once). */
	bool reserve(size_t frames, bool stereo, std::string &err);

	/* The size of the sauGenerator_run calls whose block lattice the engine reproduces (see
	 * Lattice in sau_dev_math.h): the reference starts a new <= 1024-frame block at every call
	 * (generator.c:854-878). 0 (default): every run() is one such call; otherwise a run() covers
	 * calls of that many frames each (read-ahead, file output). */
	void set_call_len(size_t frames) { call_len_ = frames > UINT32_MAX ? UINT32_MAX : (uint32_t)frames; }

	/* PCM is produced byte-swapped from the next run on (AU output) */
	void set_pcm_byteswap(bool on) { pcm_swap_ = on; }
	size_t n_streams() const { return streams_.size(); }
	Backend *backend() { return backend_; }

	/* The engine as it stands -- event positions, voice durations, operator mirrors, the call lattice's phase, and (by the
	 * backend, under `slot`) every operator's record -- so that restore() can take a render back to this frame: sauGenerator_run
	 * takes `buf_len` and `stereo` per call (sau/generator.c:905-913), and frames rendered ahead for calls of one kind are
	 * not what the reference gives a host that then asks for another. Plans are not kept: they are a function of the
	 * operator mirrors and are compiled again after a restore. */
	struct Snapshot {
		struct StreamState {
			size_t event = 0; uint32_t event_pos = 0, voice = 0; uint64_t since_event = 0;
			struct Vo { uint32_t duration, carr_op; bool init; };
			std::vector<Vo> voices;
			std::vector<OpMirror> ops;
		};
		std::vector<StreamState> streams;
		uint64_t frames_done = 0;
		uint32_t call_len = 0, lat_call = 0, call_phase = 0;
		bool out_dirty = false, valid = false;
	};
	bool snapshot(Snapshot &s, int slot, std::string &err);
	bool restore(const Snapshot &s, int slot, std::string &err);
	uint64_t frames_done() const { return frames_done_; }

private:
	struct EventNode { uint32_t wait; const sauProgramEvent *pe; };
	struct VoiceHost {
		uint32_t duration = 0;
		bool init = false;
		uint32_t carr_op = 0;
		VoicePlan plan;
		bool plan_valid = false;
		int32_t shape = -1;  /* its entry of shapes_ (voices of one shape share one step list on the device), or -1 */
	};
	/* compiled plans by graph shape (voice_plan_shape) */
	struct Shape { std::vector<uint32_t> tokens; VoicePlan plan; uint32_t plan_ofs = 0; uint64_t placed = 0; };
	std::vector<Shape> shapes_;
	std::unordered_multimap<uint64_t, uint32_t> shape_by_hash_;
	std::vector<uint32_t> shape_tokens_, shape_ids_, shape_stamp_;
	uint32_t shape_mark_ = 0;
	uint64_t rebuilds_ = 0;
	bool plan_cache_ = true, plan_check_ = false;
	bool loop_tails_ = true;
	struct Stream {
		const sauProgram *prg = nullptr;
		std::vector<EventNode> events;
		size_t event = 0;
		uint32_t event_pos = 0;
		std::vector<VoiceHost> voices;
		uint32_t voice = 0;
		float amp_scale = 0.f;
		uint32_t op_base = 0;   /* offset into the global op array */
		uint32_t vo_base = 0;
		std::vector<OpMirror> ops;
		/* per-call output length accounting (generator.c:938-949) */
		size_t call_gen = 0, part_start = 0, part_gen = 0;
		uint64_t since_event = 0; /* frames since this program's latest event */
	};

	Engine() {}
	bool handle_event(Stream &st, const EventNode &e, std::vector<OpUpdate> &batch,
			std::vector<uint8_t> &touched, std::string &err);
	bool flush_updates(std::vector<OpUpdate> &batch, std::vector<uint8_t> &touched,
			std::string &err);
	bool rebuild_plans(std::string &err);
	uint32_t estimate_sum_levels(const Stream &st, uint32_t op, uint32_t parent_dep, bool parent_varies,
			uint32_t &need, uint32_t nest) const;
	bool render_segment(uint32_t len, uint32_t offset, bool stereo, std::string &err);

	Backend *backend_ = nullptr;
	uint32_t srate_ = 0;
	std::vector<Stream> streams_;
	uint32_t total_ops_ = 0, total_voices_ = 0;
	uint32_t reserved_frames_ = 0;
	bool reserved_stereo_ = false;
	bool pcm_swap_ = false;
	bool plans_dirty_ = true;
	/* concatenated plans as uploaded; per (stream,voice) offsets */
	std::vector<Step> all_steps_;
	std::vector<FastIds> all_fast_ids_, all_fast_ids_full_;
	std::vector<uint32_t> all_op_ids_;
	struct PlanRef { uint32_t plan_ofs, plan_len, ops_ofs, nops; };
	std::vector<PlanRef> plan_refs_; /* indexed by global voice index */
	uint64_t frames_done_ = 0;
	uint32_t call_len_ = 0;          /* set_call_len() */
	uint32_t lat_call_ = 0;          /* call size of the lattice in force */
	uint32_t call_phase_ = 0;        /* frames into the reference's current call at the start of run() */
	uint32_t seg_call_pos_ = 0;      /* ... at the start of the segment being rendered */
};

/* tables.cpp */
const float *builtin_piluts();            /* 12 x 2048, built on first use */
const WaveConst *wave_consts();           /* 12 */
void override_piluts(const float *tables);/* replace (e.g. host's own tables) */

} /* namespace sauengine */
#endif
