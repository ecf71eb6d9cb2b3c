// price_f64.hip — fp64-path instantiations of the in-register pricing kernel (price_impl.hpp), the
// launch-shape rule and the launcher that picks the precision.
#include "price_impl.hpp"

namespace mcamd {

hipError_t launch_price_f32(const PathJob &j, double *d_partials, uint32_t grid, const GridFinish &fin, hipStream_t stream);
hipError_t launch_price_compact_f32(const PathJob &j, double *d_partials, unsigned long long *d_queue, uint32_t grid,
                                    const GridFinish &fin, hipStream_t stream);

// One path per thread (two in the pair-sum loop) when a path is long (fine-grained blocks keep the tail short); for short paths
// (few steps) a thread takes several, so that a block still carries a few thousand path-steps and the
// partial array stays small (1-step pricer at 100M paths: 12k partial records instead of 390k).
// Window payoffs over many paths go to the lane-compacting kernel (price_impl.hpp): a persistent grid, four
// workgroups per CU, groups of 1024 paths pulled from a queue.
uint32_t price_grid(const PathJob &j, uint32_t compute_units)
{
    if (price_compacts(j, compute_units)) return (compute_units ? compute_units : 256) * 4;
    const uint64_t np = (!j.window && j.logspace) ? kPairSumPaths : 1;   // the pair-sum loop walks this many paths per thread
    const uint64_t per_thread = j.n_sim >= 32 ? np : (32 + j.n_sim - 1) / j.n_sim * np;
    const uint64_t threads = (j.n_local + per_thread - 1) / per_thread;
    return clamp_grid((threads + kBlock - 1) / kBlock);
}

hipError_t launch_price(const PathJob &j, uint32_t compute_units, double *d_partials, unsigned long long *d_queue,
                        uint32_t grid, const FinishSpec &finish, hipStream_t stream)
{
    const GridFinish fin{finish.out, finish.ticket, finish.n_value};
    if (price_compacts(j, compute_units))
        return j.precision == 32 ? launch_price_compact_f32(j, d_partials, d_queue, grid, fin, stream)
                                 : launch_price_compact_t<double>(j, d_partials, d_queue, grid, fin, stream);
    return j.precision == 32 ? launch_price_f32(j, d_partials, grid, fin, stream)
                             : launch_price_t<double>(j, d_partials, grid, fin, stream);
}

}  // namespace mcamd
