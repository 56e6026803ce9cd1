// tuning.h -- launch-plan overrides for MEASUREMENT builds only.
//
// `make -C grafp_amd/csrc measure` compiles the same sources with -DGRAFP_MEASURE into libgrafp_hip_measure.so, which the
// tools under tools/ load through GRAFP_HIP_LIB; there GRAFP_TUNE_INT("NAME", dflt) reads the environment variable
// NAME at every call (a tool sweeps a knob inside one process by changing os.environ between launches).  In the shipping library (no -DGRAFP_MEASURE) the macro IS its default: no getenv, no mutable state, the
// plan of a launch is a pure function of its arguments (include/grafp_hip.h, "Conventions").
#pragma once
#include <cstdlib>

#ifdef GRAFP_MEASURE
namespace grafp {
inline int tune_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
}  // namespace grafp
#define GRAFP_TUNE_INT(name, dflt) (grafp::tune_env_int(name, dflt))
#else
#define GRAFP_TUNE_INT(name, dflt) (dflt)
#endif
