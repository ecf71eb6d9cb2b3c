// monte_carlo.hpp — umbrella header, as the reference's inc/monte_carlo.cuh:3-8 (which contains no
// code of its own): pulls in the whole call surface.
#pragma once

#include "BlackandScholes.hpp"
#include "tool.hpp"
#include "wrappers.hpp"
#include "testing.hpp"
