/* hip_backend.h -- the product backend: operator state in HBM, hand-written
 * gfx950 kernels (hip_backend.hip). */
#ifndef SAU_HIP_BACKEND_H
#define SAU_HIP_BACKEND_H

#include "engine.h"

namespace sauhip {

class HipBackend : public sauengine::Backend {
public:
	virtual void timing(double *render_ms, double *mix_ms, uint64_t *launches, bool reset) = 0;
	virtual void *stream_handle() = 0;
	/* 0 off, 1 time-parallel kernel only, 2 every kernel */
	virtual void set_timing(int level) = 0;
	virtual void timing_ex(double *out4, uint64_t *segments, bool reset) = 0;
	/* the next segment's rendering kernels wait for everything issued on `before` so far (sauAmd_Batch_order_after) */
	virtual bool order_after(HipBackend *before, std::string &err) = 0;
};

/* NULL (with err) when no HIP device is usable: there is no CPU fallback. device >= 0: that HIP device (sauAmd_create_Batch_on);
 * -1: the one SAU_AMD_DEVICE names, else device 0. */
HipBackend *create_hip_backend(std::string &err, int device = -1);
int device_count();
/* "domain:bus:device.function" of HIP device `dev` (bench.py: one rank per physical GPU, proved in the result line) */
bool device_pci_bus_id(int dev, char *buf, int len);

} /* namespace sauhip */
#endif
