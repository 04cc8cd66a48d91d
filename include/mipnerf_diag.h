/* mipnerf_diag.h -- C ABI of libmipnerf_diag.so: MEASUREMENT TOOLING, not part of the drop-in library.
 *
 * Built next to libmipnerf_hip.so by `python -m mipnerf_pl_amd.build` (csrc/kernels_diag.hip + csrc/diag_capi.hip).  bench.py loads it
 * when present to report, from the same process as the headline, what the chip's matrix pipe sustains under the kernels' own feeding
 * ("ceiling" in the bench line); scripts/handoff_probe.py uses the CU -> CU hand-off probe of DESIGN 4.2.  Nothing under
 * mipnerf_pl_amd/ needs it: the product path never loads this library.
 * Same conventions as mipnerf_hip.h (device pointers, stream as void*, 0 = ok, codes 1 = invalid argument / 3 = HIP error,
 * message from mipnerf_diag_last_error()). */
#ifndef MIPNERF_DIAG_H
#define MIPNERF_DIAG_H

#ifdef __cplusplus
extern "C" {
#endif

const char* mipnerf_diag_last_error(void);
/* What a back-to-back v_mfma_f32_32x32x16_bf16 stream sustains on THIS chip (the ceiling k_mlp_bf16's roofline fraction is
 * read against): 256 workgroups of waves_per_simd x 4 waves, register-resident operands (lds_reads_per_mfma = 0) or one
 * ds_read_b128 weight fragment per MFMA as in k_mlp_bf16 (1), or (2) the same plus k_mlp_bf16's weight DMA: every wave moves 8 one-KiB
 * chunks of an L2-resident 1.19-MiB stream into the LDS ring per 64 of its MFMAs with global_load_lds, or (3) that plus the training forward's
 * saved-activation stream: one 1-KiB non-temporal store per wave per 9.4 MFMAs to fresh addresses (3.7 GB per launch; 2 waves per SIMD);
 * (10) = the fp32 matrix instruction of the parity mode (v_mfma_f32_32x32x2_f32), register-fed, TFLOP/s against the 157.3 of the datasheet;
 * operands all zero or MLP-like random (weights U(-0.1,0.1),
 * activations relu(N(0,1))).  Runs for `seconds` (first half un-measured heat-up).  out3 = {TFLOP/s, ms per launch,
 * shader clock in GHz implied by the MFMA issue rate}.  Diagnostic: allocates and synchronises. */
int mipnerf_mfma_ceiling(int lds_reads_per_mfma, int waves_per_simd, int random_operands, double seconds, double* out3,
                         void* stream);
/* CU -> CU hand-off probe: 128 producer workgroups stream `tiles` tiles of tile_bytes each to 128 consumer workgroups through
 * `ring`-slot rings in global memory with counter flags (producer and consumer on the same XCD or on neighbouring XCDs;
 * store_flavour 0 = plain stores + agent release, 1 = write-through sc1 stores, 2 = plain stores, no fence, consumer loads bypass its L1
 * (same XCD only); 3 / 4 = the per-WAVE forms of 1 / 2: wave w of the producer streams sub-tiles of tile_bytes / 8 to wave w of the consumer
 * through its own ring and flags, no workgroup barrier, a whole sub-tile in flight per wave; 64 / 128 KiB tiles), mfma_per_wave register-only MFMAs per tile
 * on both sides, every word verified.  out6 = {aggregate GB/s, ms, producer stall fraction, consumer stall fraction,
 * mismatching 16-B words, 1 if a bounded poll timed out}.  Diagnostic: allocates and synchronises. */
int mipnerf_handoff_probe(int same_xcd, int store_flavour, int tiles, int ring, int tile_bytes, int mfma_per_wave, int reps,
                          double* out6, void* stream);

#ifdef __cplusplus
}
#endif
#endif
