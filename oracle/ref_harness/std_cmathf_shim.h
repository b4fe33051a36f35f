// Force-included (-include) when compiling the reference's translation unit
// with g++ 11: the reference spells a few C math functions as std::sinf,
// std::cosf, ... (hittable.h:104-105, perlin.h:18-24, texture.h:89,
// sphere.h:105-107, quad.h:41), names that libstdc++ only gained in GCC 14.
// The functions themselves exist in this image's <cmath>/libm; this header
// only makes them visible under the std:: spelling.  No arithmetic is replaced.
#pragma once
#include <cmath>
namespace std {
using ::acosf;
using ::atan2f;
using ::cosf;
using ::fabsf;
using ::floorf;
using ::fmaxf;
using ::fminf;
using ::sinf;
} // namespace std
