// oracle/ref_shim/IMU/NavState.h -- TEST INFRASTRUCTURE: stands in for src/IMU/NavState.h so that include/Frame.h compiles (the IMU code paths of
// src/Frame.cc are compiled, never executed).
#ifndef YGZ_ORACLE_REF_SHIM_NAVSTATE_H
#define YGZ_ORACLE_REF_SHIM_NAVSTATE_H
#include "frame_stubs.h"
#endif
