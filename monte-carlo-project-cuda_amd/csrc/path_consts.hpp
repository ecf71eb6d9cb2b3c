// path_consts.hpp — narrows a host-side PathJob to the per-precision kernel constants.
#pragma once

#include "launch.hpp"

#include <cmath>
#include <limits>
#include "mc_device.hpp"

namespace mcamd {

template <typename T>
inline StepConsts<T> make_consts(const PathJob &j)
{
    // exponent constants in the unit the precision's exponential consumes: log2(e) for fp32 (the step uses
    // v_exp_f32 = 2^x), 65536 / ln 2 for fp64 (f64::ExpAcc counts in 2^-16 octaves)
    const double scale = sizeof(T) == 4 ? 1.4426950408889634 : f64::kExpScale;
    StepConsts<T> c;
    c.drift = static_cast<T>(j.drift * scale);
    c.vol = static_cast<T>(j.vol * scale);
    c.vol_bm = static_cast<T>(j.vol * scale * (sizeof(T) == 4 ? 1.1774100225154747 : 1.0));  // sqrt(2 ln 2)
    c.K = static_cast<T>(j.K);
    c.B = static_cast<T>(j.B);
    c.S_start = static_cast<T>(j.S_start);
    c.logB = (j.B > 0.0 && j.S_start > 0.0) ? static_cast<T>(std::log(j.B / j.S_start) * scale)
                                             : -std::numeric_limits<T>::infinity();
    c.win_delta = sizeof(T) == 8 ? static_cast<T>(f64::exp_acc_window_delta(j.n_steps)) : T(0);
    c.P1 = j.P1;
    c.P2 = j.P2;
    c.Ik = j.Ik;
    c.n_sim = j.n_sim;
    return c;
}

inline uint32_t clamp_grid(uint64_t blocks)
{
    return static_cast<uint32_t>(blocks < 1 ? 1 : (blocks > kMaxGrid ? kMaxGrid : blocks));
}

}  // namespace mcamd
