// sdsp/filter_type.h -- MI355X engine's drop-in for simpledsp's filter_type enumeration
// (reference include/sdsp/filter_type.h:6).  The numeric values are part of the interface: they are
// the first field of the impulse-response fixtures (test/testIIR.cpp:18-19) and the `filter_type`
// argument of sdsp_hip_iir_preload() in the C ABI.
#ifndef SDSP_MI355X_FILTER_TYPE_H
#define SDSP_MI355X_FILTER_TYPE_H

namespace sdsp
{
enum class filter_type : int {
    none = 0,      // freshly constructed filter, no design yet
    low_pass = 1,  // set_lp_coeff: numerator [1, 2, 1]
    high_pass = 2, // set_hp_coeff: numerator [1, -2, 1]
    band_pass = 3, // set_bp_coeff: numerator [1, 0, -1]
    band_stop = 4  // set_bs_coeff: numerator [1, -2cos(w0), 1] (the reference's README TODO; not in its enum)
};
} // namespace sdsp

#endif // SDSP_MI355X_FILTER_TYPE_H
