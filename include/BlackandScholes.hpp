// BlackandScholes.hpp — the reference's closed-form names (inc/BlackandScholes.hpp) on top of the
// mcamd C ABI:  CND :8-30,  black_scholes_CPU :34-43.  The arithmetic (fp32 Abramowitz-Stegun
// polynomial, the reference's mixed float/double evaluation) lives in libmcamd.so and is checked
// bit for bit against the reference's own outputs (tests/golden/bs_closed_form.json).
#pragma once

#include "mcamd.h"

inline float CND(float x) { return mcamd_cnd_f32(x); }

inline void black_scholes_CPU(float &call_price, float x0, float strike_price, float T, float risk_free_rate,
                              float volatility)
{
    call_price = mcamd_bs_call_f32(x0, strike_price, T, risk_free_rate, volatility);
}
