// pgx_common.h -- internal helpers shared by the HIP translation units of libpygmu_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "pygmu_hip.h"

namespace pgx {

// Thread-local last-error message (pgx_last_error()).
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

// Library stream (created by pgx_init).  Returns nullptr before init.
hipStream_t stream();
bool initialised();

}  // namespace pgx

#define PGX_REQUIRE_INIT()                                                        \
    do {                                                                          \
        if (!pgx::initialised())                                                  \
            return pgx::fail(PGX_ERR_NOT_INIT, "pgx_init() has not been called"); \
    } while (0)

#define PGX_CHECK_ARG(cond, msg)                                     \
    do {                                                             \
        if (!(cond)) return pgx::fail(PGX_ERR_INVALID, (msg));       \
    } while (0)

#define PGX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            return pgx::fail(_e == hipErrorOutOfMemory ? PGX_ERR_NOMEM : PGX_ERR_RUNTIME, \
                             std::string(#call) + ": " + hipGetErrorString(_e));        \
        }                                                                               \
    } while (0)

// Launch-error check (does not synchronise).
#define PGX_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t _e = hipGetLastError();                                            \
        if (_e != hipSuccess)                                                         \
            return pgx::fail(PGX_ERR_RUNTIME, std::string(name) + " launch: " +       \
                                                  hipGetErrorString(_e));             \
    } while (0)

namespace pgx {

constexpr int kWave = 64;      // gfx950 wavefront
constexpr int kNumCU = 256;    // MI355X

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid size for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8.
inline int grid_for(int64_t work_items, int block) {
    int64_t g = ceil_div(work_items, block);
    if (g < 1) g = 1;
    if (g > kNumCU * 8) g = kNumCU * 8;
    return (int)g;
}

}  // namespace pgx
