// rbg_kernels_common.h -- what every kernel header of the library starts from: the C-ABI types,
// the device building blocks, the dynamic LDS of a workgroup.  Kernels and their argument structs
// live in an anonymous namespace (internal linkage): nothing but the C-ABI of include/rbg_prw.h
// leaves the translation unit (csrc/prw_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rbg_prw.h"
#include "rbg_device.h"
#include "rbg_walk.h"
#include "rbg_env.h"

namespace {

using namespace rbg;

extern __shared__ __attribute__((aligned(16))) uint32_t g_lds[];

}  // namespace
