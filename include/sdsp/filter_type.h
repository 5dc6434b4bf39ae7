// sdsp/filter_type.h -- drop-in for simpledsp's include/sdsp/filter_type.h:6.
// The enumerator values are also the first field of the impulse-response fixtures and the
// `filter_type` argument of sdsp_hip_iir_preload().
#pragma once

namespace sdsp
{
enum class filter_type : int { none = 0, low_pass = 1, high_pass = 2, band_pass = 3 };
}
