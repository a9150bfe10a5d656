// placeholder until the fused read convolver lands
#include "kernels.h"
namespace hello {
int readconv_reads_per_group() { return 8; }
hipError_t launch_readconv_fused(const ReadConvArgs&, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_readconv_finalize(const float*, const int32_t*, float*, int, hipStream_t) { return hipErrorNotSupported; }
}
