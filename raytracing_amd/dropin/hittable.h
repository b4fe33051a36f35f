// hittable.h of the reference API; everything lives in rtamd_scene_api.hpp.
#pragma once
#include "rtamd_scene_api.hpp"
