/* program_io.cpp -- sauProgram <-> relocatable image.
 *
 * Image layout (little endian, 8-byte aligned blocks):
 *   [0]  "SAUPIMG1"            8 bytes
 *   [8]  total size            u64
 *   [16] sauProgram            (pointers hold offsets from image start, 0 = NULL)
 *   ...  sauProgramEvent[ev_count], then per event its sauProgramOpRef[] and
 *        sauProgramOpData[], then per op-data its sauLine and sauProgramIDArr
 *        blocks (identical id arrays are stored once).
 * The structs are the ABI structs of include/sau_abi.h, so loading is one
 * memcpy plus pointer fix-ups.
 */
#include "../../include/saugns_amd.h"
#include <exception>
#include <map>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

const char MAGIC[8] = {'S', 'A', 'U', 'P', 'I', 'M', 'G', '1'};

struct Writer {
	std::vector<uint8_t> bytes;
	std::map<const void *, uint64_t> seen;
	uint64_t put(const void *src, size_t n) {
		size_t at = (bytes.size() + 7) & ~(size_t)7;
		bytes.resize(at + n);
		memcpy(&bytes[at], src, n);
		return at;
	}
	uint64_t put_line(const sauLine *l) {
		if (!l) return 0;
		auto it = seen.find(l);
		if (it != seen.end()) return it->second;
		uint64_t at = put(l, sizeof(sauLine));
		seen[l] = at;
		return at;
	}
	uint64_t put_ids(const sauProgramIDArr *a) {
		if (!a) return 0;
		auto it = seen.find(a);
		if (it != seen.end()) return it->second;
		uint64_t at = put(a, sizeof(uint32_t) * (1 + (size_t)a->count));
		seen[a] = at;
		return at;
	}
};

template <typename T> void set_ptr(T *&field, uint64_t off) {
	field = (T *)(uintptr_t)off;
}

} /* namespace */

static size_t serialize(const sauProgram *prg, void *buf, size_t cap);
extern "C" size_t sauAmd_program_serialize(const sauProgram *prg, void *buf, size_t cap) {
	if (!prg) return 0;
	try { return serialize(prg, buf, cap); } catch (const std::exception &) { return 0; } /* (nothing C++ crosses the C ABI) */
}
static size_t serialize(const sauProgram *prg, void *buf, size_t cap) {
	Writer w;
	w.bytes.resize(16, 0);
	memcpy(&w.bytes[0], MAGIC, 8);
	sauProgram hdr = *prg;
	hdr.name = nullptr; hdr.mp = nullptr; hdr.parse = nullptr;
	uint64_t prg_at = w.put(&hdr, sizeof hdr);
	std::vector<sauProgramEvent> evs(prg->events, prg->events + prg->ev_count);
	for (size_t i = 0; i < evs.size(); ++i) {
		const sauProgramEvent &src = prg->events[i];
		uint64_t list_at = 0, data_at = 0;
		if (src.op_list && src.op_count)
			list_at = w.put(src.op_list, sizeof(sauProgramOpRef) * src.op_count);
		if (src.op_data && src.op_data_count) {
			std::vector<sauProgramOpData> ods(src.op_data, src.op_data + src.op_data_count);
			for (sauProgramOpData &od : ods) {
				set_ptr(od.pan, w.put_line(od.pan));
				set_ptr(od.amp, w.put_line(od.amp));
				set_ptr(od.amp2, w.put_line(od.amp2));
				set_ptr(od.freq, w.put_line(od.freq));
				set_ptr(od.freq2, w.put_line(od.freq2));
				set_ptr(od.pm_a, w.put_line(od.pm_a));
				set_ptr(od.camods, w.put_ids(od.camods));
				set_ptr(od.amods, w.put_ids(od.amods));
				set_ptr(od.ramods, w.put_ids(od.ramods));
				set_ptr(od.fmods, w.put_ids(od.fmods));
				set_ptr(od.rfmods, w.put_ids(od.rfmods));
				set_ptr(od.pmods, w.put_ids(od.pmods));
				set_ptr(od.apmods, w.put_ids(od.apmods));
				set_ptr(od.fpmods, w.put_ids(od.fpmods));
			}
			data_at = w.put(ods.data(), sizeof(sauProgramOpData) * ods.size());
		}
		set_ptr(evs[i].op_list, list_at);
		set_ptr(evs[i].op_data, data_at);
	}
	uint64_t ev_at = evs.empty() ? 0 : w.put(evs.data(), sizeof(sauProgramEvent) * evs.size());
	sauProgram *out_hdr = (sauProgram *)&w.bytes[prg_at];
	set_ptr(out_hdr->events, ev_at);
	uint64_t total = w.bytes.size();
	memcpy(&w.bytes[8], &total, 8);
	if (buf && cap >= w.bytes.size())
		memcpy(buf, w.bytes.data(), w.bytes.size());
	return w.bytes.size();
}

namespace {
/* An offset field becomes a pointer to `count` objects of `size` bytes, all inside the image:
 * 8-byte aligned (as the writer lays blocks out), past the header, no overflow in the product;
 * 0 is NULL and only allowed when nothing is counted there. */
template <typename T> bool fix(T *&field, uint8_t *base, size_t len, size_t size, size_t count,
		bool null_ok_when_empty = true) {
	uint64_t off = (uint64_t)(uintptr_t)field;
	if (off == 0) { field = nullptr; return count == 0 && null_ok_when_empty; }
	if (off % 8 != 0 || off < 16 + sizeof(sauProgram) || off > len) return false;
	if (size && count > (len - off) / size) return false;
	field = (T *)(base + off);
	return true;
}
} /* namespace */

extern "C" sauProgram *sauAmd_program_load(const void *image, size_t len) {
	if (!image || len < 16 + sizeof(sauProgram) || memcmp(image, MAGIC, 8) != 0)
		return nullptr;
	uint64_t total;
	memcpy(&total, (const uint8_t *)image + 8, 8);
	if (total > len || total < 16 + sizeof(sauProgram)) return nullptr;
	uint8_t *base = (uint8_t *)malloc(total);
	if (!base) return nullptr;
	memcpy(base, image, total);
	sauProgram *prg = (sauProgram *)(base + 16);
	bool ok = fix(prg->events, base, total, sizeof(sauProgramEvent), prg->ev_count);
	uint64_t max_id = 0; /* the highest operator id anything in the image names */
	for (size_t i = 0; ok && i < prg->ev_count; ++i) {
		sauProgramEvent *ev = (sauProgramEvent *)&prg->events[i];
		if (ev->op_list == nullptr) ev->op_count = 0; /* (the list is optional: generator.c never reads it) */
		ok = ok && fix(ev->op_list, base, total, sizeof(sauProgramOpRef), ev->op_count);
		ok = ok && fix(ev->op_data, base, total, sizeof(sauProgramOpData), ev->op_data_count);
		for (size_t k = 0; ok && ev->op_list && k < ev->op_count; ++k)
			if (ev->op_list[k].id > max_id) max_id = ev->op_list[k].id;
		if (ok && ev->carr_op_id > max_id && ev->vo_id != SAU_PVO_NO_ID) max_id = ev->carr_op_id;
		for (size_t k = 0; ok && k < ev->op_data_count; ++k) {
			sauProgramOpData *od = (sauProgramOpData *)&ev->op_data[k];
			if (od->id > max_id) max_id = od->id;
			sauLine **lines[] = {&od->pan, &od->amp, &od->amp2, &od->freq, &od->freq2, &od->pm_a};
			for (sauLine **l : lines) ok = ok && (*l == nullptr || fix(*l, base, total, sizeof(sauLine), 1));
			const sauProgramIDArr **arrs[] = {&od->camods, &od->amods, &od->ramods, &od->fmods,
				&od->rfmods, &od->pmods, &od->apmods, &od->fpmods};
			for (const sauProgramIDArr **a : arrs) {
				if (*a == nullptr) continue;
				ok = ok && fix(*a, base, total, sizeof(uint32_t), 1);
				if (ok) {
					const size_t at = (size_t)((const uint8_t *)(*a) - base);
					ok = (*a)->count <= (total - at) / sizeof(uint32_t) - 1;
					for (uint32_t j = 0; ok && j < (*a)->count; ++j)
						if ((*a)->ids[j] > max_id) max_id = (*a)->ids[j];
				}
			}
		}
	}
	/* The engine allocates per-operator state for op_count operators: the count is bounded by what the image can name --
	 * the highest operator id in any operator datum, graph list, modulator list or carrier field, plus the ids the parser
	 * may have counted without ever using them (devtests/freelist.sau: 4 ids, 2 with data, the others named nowhere; 64 of
	 * slack) -- so a forged count cannot make a small image cost gigabytes (until round 5: op_count <= bytes).
	 * (vo_count is 16 bits wide.) */
	ok = ok && prg->op_count <= max_id + 1 + 64 && prg->op_count <= total / sizeof(sauProgramOpRef);
	if (!ok) { free(base); return nullptr; }
	prg->name = "image";
	return prg;
}

extern "C" void sauAmd_program_free(sauProgram *prg) {
	if (prg) free((uint8_t *)prg - 16);
}
