// maray_libm.h — PLACEHOLDER (first bring-up only): OCML sin/exp/log.
// Replaced by the bit-exact glibc 2.35 port; chess parity does not depend on it
// because every Sin of that scene feeds a Step (only the sign is observable).
#pragma once
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ double maray_libm_sin(double a) { return ::sin(a); }
__device__ __forceinline__ double maray_libm_exp(double a) { return ::exp(a); }
__device__ __forceinline__ double maray_libm_log(double a) { return ::log(a); }
#endif
