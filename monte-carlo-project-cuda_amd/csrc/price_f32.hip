// price_f32.hip — fp32-path instantiations of the in-register pricing kernel (price_impl.hpp).
#include "price_impl.hpp"

namespace mcamd {

hipError_t launch_price_f32(const PathJob &j, double *d_partials, uint32_t grid, const GridFinish &fin, hipStream_t stream)
{
    return launch_price_t<float>(j, d_partials, grid, fin, stream);
}

hipError_t launch_price_compact_f32(const PathJob &j, double *d_partials, unsigned long long *d_queue, uint32_t grid,
                                    const GridFinish &fin, hipStream_t stream)
{
    return launch_price_compact_t<float>(j, d_partials, d_queue, grid, fin, stream);
}

}  // namespace mcamd
