// price_f64.hip — fp64-path instantiations of the in-register pricing kernel (price_impl.hpp), the
// launch-shape rule and the launcher that picks the precision.
#include "price_impl.hpp"

namespace mcamd {

hipError_t launch_price_f32(const PathJob &j, double *d_partials, uint32_t grid, hipStream_t stream);

// One path per thread when a path is long (fine-grained blocks keep the tail short); for short paths
// (few steps) a thread takes several, so that a block still carries a few thousand path-steps and the
// partial array stays small (1-step pricer at 100M paths: 12k partial records instead of 390k).
uint32_t price_grid(uint64_t n_local, uint32_t n_sim)
{
    const uint64_t per_thread = n_sim >= 32 ? 1 : (32 + n_sim - 1) / n_sim;
    const uint64_t threads = (n_local + per_thread - 1) / per_thread;
    return clamp_grid((threads + kBlock - 1) / kBlock);
}

hipError_t launch_price(const PathJob &j, double *d_partials, uint32_t grid, hipStream_t stream)
{
    return j.precision == 32 ? launch_price_f32(j, d_partials, grid, stream)
                             : launch_price_t<double>(j, d_partials, grid, stream);
}

}  // namespace mcamd
