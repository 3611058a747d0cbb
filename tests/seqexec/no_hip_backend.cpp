/* Test infrastructure: the device backend's entry points for the sanitizer build of the host control plane
 * (tests/seqexec/Makefile `asan`). That build holds engine.cpp, plan.cpp, capi.cpp, program_io.cpp, bank_builder.cpp,
 * sndout.cpp and tables.cpp as they ship, compiled by g++ with -fsanitize=address,undefined, and no HIP: every path
 * that would need a device reports that there is none -- exactly what the product does on a box without a GPU (no CPU
 * fallback) -- and the tests drive the control plane through the sequential executor (seq_backend.cpp) instead. */
#include "../../saugns_amd/csrc/hip_backend.h"

namespace sauhip {
HipBackend *create_hip_backend(std::string &err, int) { err = "no HIP device (sanitizer build of the host control plane: there is no device backend in it)"; return nullptr; }
int device_count() { return 0; }
bool device_pci_bus_id(int, char *, int) { return false; }
} /* namespace sauhip */
