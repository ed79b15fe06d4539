// Minimal stand-in for <boost/version.hpp> (oracle build only).
// A value below 106100 makes util/small_vector.h take its std::vector fallback.
#pragma once
#define BOOST_VERSION 105700
