/* capi_internal.h -- what the test-hook library (tests/hooks/test_hooks.cpp -> tests/hooks/libsaugns_amd_hooks.so) reaches
 * of capi.cpp and sndout.cpp beside the C ABI. Hidden visibility: none of this is an export of libsaugns_amd.so. The hook
 * library links the product's own objects and adds the entry points tests/ use to run the host control plane over an
 * injected backend (tests/seqexec) and to probe the device-compiled arithmetic; the product library has none of them
 * (VERDICT r04 item 9). */
#ifndef SAU_CAPI_INTERNAL_H
#define SAU_CAPI_INTERNAL_H
#include "../../include/saugns_amd.h"
#include "engine.h"

namespace sauamd_internal {
/* sau_create_Generator / sauAmd_create_Batch over a caller-supplied backend (owned by the object made; NULL: the HIP one) */
sauGenerator *make_generator(const sauProgram *prg, uint32_t srate, sauengine::Backend *injected);
sauAmdBatch *make_batch_over(const sauProgram *const *prgs, size_t n, uint32_t srate, sauengine::Backend *injected, int device = -1);
/* how often a call of another size or channel layout took this generator's read-ahead back (capi.cpp: generator_rewind) */
unsigned generator_rewinds(const sauGenerator *g);
/* sauAmd_render_file's body (sndout.cpp) */
bool render_file(const sauProgram *prg, uint32_t srate, const char *path, int format, int channels,
		sauengine::Backend *injected, uint64_t *frames_out, std::string &err);
} /* namespace sauamd_internal */
#endif
