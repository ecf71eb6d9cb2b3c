// option_price.hpp — the reference's inc/option_price.hpp:1-6 is an empty stub (two includes and a
// using-directive, included by nobody).  The name is kept as the public entry header of the engine.
#pragma once

#include "monte_carlo.hpp"
