// C ABI of libmipnerf_diag.so (include/mipnerf_diag.h): the in-process MFMA ceilings and the CU -> CU hand-off probe.
// Measurement tooling kept OUT of the drop-in library (VERDICT r03 hygiene): its kernels live in kernels_diag.hip.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <string>

#include "../../include/mipnerf_diag.h"
#include "kernels.hpp"

namespace {

thread_local std::string g_err;
enum { DIAG_OK = 0, DIAG_E_INVALID = 1, DIAG_E_HIP = 3 };      // the codes of mipnerf_hip.h

int fail(int code, const char* msg) {
    g_err = msg;
    return code;
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" {

const char* mipnerf_diag_last_error(void) { return g_err.c_str(); }

int mipnerf_mfma_ceiling(int lds_reads_per_mfma, int waves_per_simd, int random_operands, double seconds, double* out3, void* stream) {
    if (!out3 || waves_per_simd < 1 || waves_per_simd > 2 || lds_reads_per_mfma < 0 || (lds_reads_per_mfma > 3 && lds_reads_per_mfma != 10) || !(seconds > 0) || seconds > 30 ||
        (lds_reads_per_mfma == 3 && waves_per_simd != 2))
        return fail(DIAG_E_INVALID, "mfma_ceiling: waves_per_simd in {1,2}, lds_reads_per_mfma (feeding mode) in {0,1,2,3,10; 3 needs 2 waves per SIMD}, 0 < seconds <= 30");
    char msg[256];
    msg[0] = 0;
    const int rc = mip::run_mfma_ceiling(lds_reads_per_mfma, waves_per_simd, random_operands, seconds, out3, out3 + 1, out3 + 2, S(stream), msg, sizeof msg);
    if (rc != 0) g_err = msg;       // a successful call leaves the last error alone
    return rc == 0 ? DIAG_OK : DIAG_E_HIP;
}

int mipnerf_handoff_probe(int same_xcd, int store_flavour, int tiles, int ring, int tile_bytes, int mfma_per_wave, int reps, double* out6,
                          void* stream) {
    if (!out6 || tiles < 1 || tiles > (1 << 16) || ring < 1 || ring > 64 || tile_bytes < 65536 || tile_bytes > (1 << 22) || reps < 1 || reps > 16 ||
        mfma_per_wave < 0 || mfma_per_wave > 4096 || (long long)ring * tile_bytes * 128 > (4ll << 30) || store_flavour < 0 || store_flavour > 4 ||
        ((store_flavour == 2 || store_flavour == 4) && !same_xcd) || (store_flavour >= 3 && tile_bytes != 65536 && tile_bytes != 131072))
        return fail(DIAG_E_INVALID, "handoff_probe: argument out of range (flavours 0-4; 2 / 4 are same-XCD protocols; 3 / 4 take 64 / 128 KiB tiles)");
    char msg[256];
    msg[0] = 0;
    const int rc = mip::run_handoff_probe(same_xcd, store_flavour, tiles, ring, tile_bytes, mfma_per_wave, reps, out6, S(stream), msg, sizeof msg);
    if (rc != 0) g_err = msg;
    return rc == 0 ? DIAG_OK : (rc == -2 ? DIAG_E_INVALID : DIAG_E_HIP);
}

}  // extern "C"
