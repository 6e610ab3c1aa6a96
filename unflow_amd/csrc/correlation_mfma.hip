// placeholder — replaced by the MFMA banded-Gram kernels
#include "common.h"
#include "correlation_geom.h"
int corr_mfma_supported(const CorrGeom&, int, int) { return 0; }
int corr_mfma_fwd(const float*, const float*, int, int, float*, int, int, int, int, int, const CorrGeom&, hipStream_t) { return UNFLOW_ERR_UNSUPPORTED; }
int corr_mfma_bwd(const float*, int, const float*, const float*, int, int, float*, float*, int, int, int, int, int, int, const CorrGeom&, hipStream_t) { return UNFLOW_ERR_UNSUPPORTED; }
