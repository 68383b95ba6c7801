/*
 * sperr_hip.h -- C ABI of libsperr_hip.so, the MI355X (gfx950) SPERR 3D chunk-pipeline engine.
 *
 * Two families of entry points:
 *
 * 1. Drop-in replacements for the reference's C API (same names, argument meaning, ownership
 *    and return codes as /root/reference/include/SPERR_C_API.h:106-137,87-92 and
 *    /root/reference/src/SPERR_C_API.cpp:135-258): host buffers in, malloc'd host buffers out.
 *    The per-chunk pipeline (conditioner, CDF 9/7 DWT, quantiser, SPECK3D coder, bit packing)
 *    runs on the GPU(s); `nthreads` (the reference's OpenMP team size, src/SPERR3D_OMP_C.cpp:12-20)
 *    is the number of host threads that stage rows for the transfers.  mode 1 (fixed rate, `quality` = bits per value), mode 2
 *    (fixed PSNR, dB) and mode 3 (fixed point-wise error, the tolerance; the outlier list goes
 *    through the reference's Outlier_Coder / SPECK1D_INT stream format) are implemented.
 *
 * 2. Device-resident entry points (sperrhip_*): the volume and the container stay in HBM, which
 *    is what bench.py times and what an application that already holds its field on the GPU
 *    should call.  All pointers marked `d_` are device pointers; `hip_stream` is a hipStream_t
 *    passed as void* (NULL = the default stream).  Calls are synchronous with respect to the
 *    host when they return.
 *
 * There is no CPU fallback: every entry point returns -1 (and prints the HIP error) when no
 * gfx950 device or kernel image is available.
 */
#ifndef SPERR_HIP_H
#define SPERR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference-compatible C API (replaces include/SPERR_C_API.h of the reference) ---------- */

/* include/SPERR_C_API.h:106-119 ; returns 0 ok, 1 *dst not NULL, 2 bad parameter, -1 other */
int sperr_comp_3d(const void* src, int is_float, size_t dimx, size_t dimy, size_t dimz,
                  size_t chunk_x, size_t chunk_y, size_t chunk_z, int mode, double quality,
                  size_t nthreads, void** dst, size_t* dst_len);

/* include/SPERR_C_API.h:129-137 */
int sperr_decomp_3d(const void* src, size_t src_len, int output_float, size_t nthreads,
                    size_t* dimx, size_t* dimy, size_t* dimz, void** dst);

/* include/SPERR_C_API.h:87-92 */
void sperr_parse_header(const void* src, size_t* dimx, size_t* dimy, size_t* dimz, int* is_float);

/* include/SPERR_C_API.h:138-156 : keep `pct` percent of every chunk stream (progressive access);
 * the result decodes with sperr_decomp_3d.  Host only.  Returns 0 ok, 1 *dst not NULL, -1 other. */
int sperr_trunc_3d(const void* src, size_t src_len, unsigned pct, void** dst, size_t* dst_len);

/* ---- chunk farm: the same on an explicit list of devices -----------------------------------
 * sperr_comp_3d / sperr_decomp_3d above run on the chunk farm of sperr_amd/csrc/farm.hip: the
 * volume is cut into work items (runs of equally shaped chunks in chunk_volume order,
 * /root/reference/src/sperr_helper.cpp:542-592), a shared queue hands them to a few worker threads
 * per device, and every worker streams its items through pinned staging buffers (gather -> H2D ->
 * the device pipeline -> D2H), so the volume is never resident on a device as a whole and may be
 * larger than HBM.  This replaces the OpenMP chunk loops of
 * /root/reference/src/SPERR3D_OMP_C.cpp:94-130 and src/SPERR3D_OMP_D.cpp:101-127.
 * They use the devices SPERR_HIP_DEVICES names ("all", the default, or "0,2,3"); the entry points
 * below take the list as an argument (devices == NULL or ndevices == 0: the same default).  An
 * ordinal may repeat.  `nthreads`: host threads that move rows between the caller's buffers and the
 * staging buffers (0 = a default of 4 to 12 per worker, by the host's thread count).  A caller's volume that is pinned host memory
 * (hipHostMalloc / hipHostRegister) is read and written by DMA without staging; a volume that is
 * DEVICE memory (compress: src, decompress: the dst of sperrhip_decomp_3d_into) is handled by the
 * device-resident path on the device it lives on, and only the container crosses the bus.
 * Return codes as sperr_comp_3d / sperr_decomp_3d. */
int sperrhip_comp_3d_farm(const void* src, int is_float, size_t dimx, size_t dimy, size_t dimz,
                          size_t chunk_x, size_t chunk_y, size_t chunk_z, int mode, double quality,
                          size_t nthreads, const int* devices, size_t ndevices, void** dst,
                          size_t* dst_len);
int sperrhip_decomp_3d_farm(const void* src, size_t src_len, int output_float, size_t nthreads,
                            const int* devices, size_t ndevices, size_t* dimx, size_t* dimy,
                            size_t* dimz, void** dst);
/* the volume into a buffer of the caller (dst_bytes of room; it may be pinned memory) */
int sperrhip_decomp_3d_into(const void* src, size_t src_len, int output_float, size_t nthreads,
                            const int* devices, size_t ndevices, void* dst, size_t dst_bytes,
                            size_t* dimx, size_t* dimy, size_t* dimz);
/* Host-only check of the farm's queue (no device is touched): how the chunks of a volume are cut
 * into items and which of ndevices * workers_per_device idle workers takes which.  lockstep != 0:
 * the workers take their items in rounds (all equally fast), which makes the outcome
 * deterministic.  per_worker_chunks[ndevices * workers_per_device], item_of_chunk[nchunks],
 * worker_of_chunk[nchunks] (each may be NULL); worker w belongs to device w mod ndevices. */
int sperrhip_farm_selftest(size_t dimx, size_t dimy, size_t dimz, size_t chunk_x, size_t chunk_y,
                           size_t chunk_z, size_t bytes_per_value, size_t ndevices,
                           size_t workers_per_device, int lockstep, uint32_t* per_worker_chunks,
                           uint32_t* item_of_chunk, uint32_t* worker_of_chunk, size_t* nitems);
/* NUMA placement of the farm.  A worker thread binds itself (its helper threads inherit the mask,
 * the staging memory it pins is first touched by it) to the CPUs of the NUMA node its device hangs
 * off: hipDeviceGetPCIBusId -> <sysfs>/bus/pci/devices/<bdf>/numa_node ->
 * <sysfs>/devices/system/node/node<N>/cpulist.  The reference leaves placement to the OpenMP runtime
 * of its chunk loop (/root/reference/src/SPERR3D_OMP_C.cpp:94-130, SPERR3D_OMP_D.cpp:101-127).
 * SPERR_HIP_FARM_NUMA=0 switches it off, SPERR_HIP_SYSFS_ROOT names another tree.  The first two
 * are host only (no device is touched; sysfs_root NULL = the default tree): what the tree says
 * about a PCI device, and the binding applied to the CALLING thread (returns the number of CPUs it
 * is bound to; 0 = left as it was).  The third reports a device's PCI address, node (-1 unknown)
 * and CPU count as the farm sees them. */
int sperrhip_numa_probe(const char* sysfs_root, const char* pci_bdf, int* node, int* cpus,
                        size_t cpus_cap, size_t* ncpus);
int sperrhip_numa_bind_self(const char* sysfs_root, const char* pci_bdf);
int sperrhip_farm_device_place(int dev, char* bdf, size_t bdf_cap, int* node, size_t* ncpus);
/* Host CPUs this process may use, and what the farm does with them (host only, no device touched).
 * The reference sizes its chunk loop by the caller's nthreads, 0 = omp_get_max_threads()
 * (/root/reference/src/SPERR3D_OMP_C.cpp:12-20).  The farm starts threads of its own, and sizes them by
 * min(affinity mask, cgroup CFS quota) -- cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us, the
 * tightest over the process's group and its ancestors -- never by the machine's CPU count: a container
 * that shows 256 CPUs and grants 16 throttles what runs beyond 16.  cgroup_root / proc_cgroup NULL = the
 * system's (SPERR_HIP_CGROUP_ROOT / SPERR_HIP_PROC_CGROUP name others).  quota_cpus 0 = unlimited.
 * sperrhip_host_throttle reads cpu.stat (returns 1 when there is none).  sperrhip_farm_threads reports
 * the thread plan for ndevices devices. */
int sperrhip_host_cpus(const char* cgroup_root, const char* proc_cgroup, size_t* visible, size_t* affinity,
                       double* quota_cpus, size_t* usable);
int sperrhip_host_throttle(const char* cgroup_root, const char* proc_cgroup, unsigned long long* nr_throttled,
                           unsigned long long* throttled_usec);
int sperrhip_farm_threads(size_t nthreads, size_t ndevices, size_t* workers_per_device,
                          size_t* dec_workers_per_device, size_t* helpers_per_worker, size_t* threads_total,
                          size_t* cpus_usable);

/* ---- device-resident API ---------------------------------------------------------------------- */

/* Upper bound of the container size sperrhip_compress_dev can produce (bytes). */
size_t sperrhip_max_compressed_size(size_t dimx, size_t dimy, size_t dimz, size_t chunk_x,
                                    size_t chunk_y, size_t chunk_z, int mode, double quality);

/* Compress a volume that lives in device memory (x fastest; float if is_float else double) into
 * a SPERR container written to d_dst (device memory, capacity dst_cap bytes).  *dst_len receives
 * the container length.  Same modes / return codes as sperr_comp_3d (1 is never returned). */
int sperrhip_compress_dev(const void* d_src, int is_float, size_t dimx, size_t dimy, size_t dimz,
                          size_t chunk_x, size_t chunk_y, size_t chunk_z, int mode, double quality,
                          void* d_dst, size_t dst_cap, size_t* dst_len, void* hip_stream);

/* Decompress a container that lives in device memory into d_dst (device memory holding
 * dimx*dimy*dimz floats or doubles; query the dims first with sperrhip_parse_header_dev or keep
 * them from compression).  Returns 0 ok, -1 error. */
int sperrhip_decompress_dev(const void* d_src, size_t src_len, int output_float, void* d_dst,
                            size_t dst_cap_bytes, size_t* dimx, size_t* dimy, size_t* dimz,
                            void* hip_stream);

/* Reads the container header of a device-resident container. Returns 0 ok. */
int sperrhip_parse_header_dev(const void* d_src, size_t src_len, size_t* dimx, size_t* dimy,
                              size_t* dimz, int* is_float, size_t* chunk_x, size_t* chunk_y,
                              size_t* chunk_z);

/* ---- stage access for parity tests (device pointers; one chunk = the whole array) ------------- */

/* In-place forward / inverse CDF 9/7 3D transform of dimx*dimy*dimz doubles. */
int sperrhip_dwt3d_dev(double* d_vals, size_t dimx, size_t dimy, size_t dimz, int inverse,
                       void* hip_stream);

/* SPECK3D-encode one chunk of already quantised coefficients.  width is 4 or 8 (bytes per
 * magnitude); d_sign holds (n+63)/64 words, bit i set = non-negative.  budget_bits 0 = unlimited.
 * The 9-byte-header stream is written to d_dst; *dst_len receives its length. */
int sperrhip_speck3d_encode_dev(const void* d_coef, int width, const uint64_t* d_sign, size_t dimx,
                                size_t dimy, size_t dimz, size_t budget_bits, void* d_dst,
                                size_t dst_cap, size_t* dst_len, void* hip_stream);

/* Inverse of the above: d_coef (width 4 if the header says <= 32 planes, else 8) and d_sign are
 * outputs. *width_out receives the width that was written. */
int sperrhip_speck3d_decode_dev(const void* d_stream, size_t stream_len, size_t dimx, size_t dimy,
                                size_t dimz, void* d_coef, uint64_t* d_sign, int* width_out,
                                void* hip_stream);

/* ---- 2D slices -------------------------------------------------------------------------------
 * Drop-in replacements for /root/reference/include/SPERR_C_API.h:53-81
 * (src/SPERR_C_API.cpp:7-134): one slice through SPECK2D_FLT (dwt2d, SPECK2D_INT), the three
 * modes of sperr_comp_3d; out_inc_header != 0 prepends the 10-byte header {version, flags,
 * u32 dimx, u32 dimy}; sperr_decomp_2d takes the stream WITHOUT that header. */
int sperr_comp_2d(const void* src, int is_float, size_t dimx, size_t dimy, int mode, double quality,
                  int out_inc_header, void** dst, size_t* dst_len);
int sperr_decomp_2d(const void* src, size_t src_len, int output_float, size_t dimx, size_t dimy,
                    void** dst);
/* the same on device-resident buffers */
size_t sperrhip_max_compressed_size_2d(size_t dimx, size_t dimy, int mode, double quality);
int sperrhip_compress_2d_dev(const void* d_src, int is_float, size_t dimx, size_t dimy, int mode,
                             double quality, int out_inc_header, void* d_dst, size_t dst_cap,
                             size_t* dst_len, void* hip_stream);
int sperrhip_decompress_2d_dev(const void* d_src, size_t src_len, int output_float, size_t dimx,
                               size_t dimy, void* d_dst, size_t dst_cap_bytes, void* hip_stream);

/* ---- multi-resolution decoding ---------------------------------------------------------------
 * sperr::SPERR3D_OMP_D::decompress(p, multi_res = true) + release_hierarchy()
 * (/root/reference/include/SPERR3D_OMP_D.h:22-29, src/SPERR3D_OMP_D.cpp:50-150,
 * src/CDF97.cpp:150-168, src/SPECK_FLT.cpp:592-603): besides the volume, the volume at every
 * coarsened resolution of the chunks (src/sperr_helper.cpp:70-123), coarsest first, as doubles.
 * Exists only when the chunks are dyadic and tile the volume; otherwise there are 0 levels. */
/* number of levels and their dims (level_dims: 3 entries per level, x y z; room for 16 levels) */
int sperrhip_multires_levels(size_t dimx, size_t dimy, size_t dimz, size_t chunk_x, size_t chunk_y,
                             size_t chunk_z, size_t* nlev, size_t* level_dims);
/* d_levels: host array of `nlev` device pointers, level h sized by sperrhip_multires_levels */
int sperrhip_decompress_multires_dev(const void* d_src, size_t src_len, int output_float,
                                     void* d_dst, size_t dst_cap_bytes, size_t nlev,
                                     double* const* d_levels, void* hip_stream);
/* host buffers: *dst must be NULL; *dst and levels[0 .. *nlev) are malloc'd (free() them) */
int sperrhip_decomp_3d_multires(const void* src, size_t src_len, int output_float, size_t* dimx,
                                size_t* dimy, size_t* dimz, void** dst, size_t* nlev,
                                size_t* level_dims, double** levels);

/* The same for a slice: SPECK2D_FLT::decompress(multi_res = true) + release_hierarchy()
 * (/root/reference/src/SPECK2D_FLT.cpp:52-58, src/CDF97.cpp:114-130,566-579,
 * src/SPECK_FLT.cpp:592-603; what utilities/sperr2d.cpp --decomp_lowres_f/_d writes): the slice at
 * every coarsened resolution (src/sperr_helper.cpp:86-95), coarsest first, as doubles.  Every
 * slice shape has them (none if the shorter side is below 8).  level_dims: 2 entries per level
 * (x y), room for 16 levels.  `src` is the stream WITHOUT the 10-byte header, as for
 * sperr_decomp_2d. */
int sperrhip_multires_levels_2d(size_t dimx, size_t dimy, size_t* nlev, size_t* level_dims);
int sperrhip_decompress_2d_multires_dev(const void* d_src, size_t src_len, int output_float,
                                        size_t dimx, size_t dimy, void* d_dst, size_t dst_cap_bytes,
                                        size_t nlev, double* const* d_levels, void* hip_stream);
/* host buffers: *dst must be NULL; *dst and levels[0 .. *nlev) are malloc'd (free() them) */
int sperrhip_decomp_2d_multires(const void* src, size_t src_len, int output_float, size_t dimx,
                                size_t dimy, void** dst, size_t* nlev, size_t* level_dims,
                                double** levels);

/* ---- profiling ------------------------------------------------------------------------------ */

/* When enabled, the engine brackets every pipeline stage with HIP events on the launch stream
 * and accumulates their durations. */
void sperrhip_profile_enable(int on);
/* Restrict the bracketing to one kernel (a name sperrhip_profile_get reported; NULL or "" = all):
 * two event records per launch of every kernel cost a few percent of a step. */
void sperrhip_profile_only(const char* kernel);
void sperrhip_profile_reset(void);
/* Fills up to `cap` entries; returns the number of stages. names[i] points to a static string.
 * millis: time during which the kernel was running (decoding enqueues sub-batches on several
 * streams, whose launches of one kernel overlap; for everything else this is the sum of the
 * launch durations). */
int sperrhip_profile_get(const char** names, double* millis, int* launches, int cap);
/* Same, plus the plain sum of the launch durations (sum / launches = what a kernel trace
 * reports as the average duration); sum_millis may be NULL. */
int sperrhip_profile_get2(const char** names, double* busy_millis, double* sum_millis,
                          int* launches, int cap);

/* Diagnostics of the table-driven LIS decoder: when `on`, thread 0 of every decoding workgroup
 * accumulates shader-clock ticks per phase; out64 (64 entries, may be NULL) receives the counters
 * of chunk 0 of the last decoded batch: [0..10] load, tables, hopS, P1, P2, P3, P4, expand,
 * compaction, windows, placement; [16+K] windows, [32+K] table ticks and [48+K] stream bits of
 * the list levels whose class chain has length K. */
void sperrhip_debug_lis_stamps(int on, unsigned long long* out64);

/* Diagnostics: how often this process took one of the workspace-saving paths (for the tests that
 * have to know they ran): 0 = 64-bit retries of a batch whose coder arrays lay over the chunk buffer
 * (the batch is transformed again), 1 = compression batches with the coder arrays over the chunk
 * buffer, 2 = decompression batches with a compact chunk buffer, 3 = bytes of the largest workspace
 * arena an engine of this process holds right now, 4 / 5 = bytes of pinned staging memory / of device
 * buffers the chunk farm's workers hold right now (all devices).  Other values: 0. */
unsigned long long sperrhip_debug_counter(int which);

/* Gives back what the library keeps between calls (it keeps workspaces, shape tables, pinned staging
 * buffers and device buffers so that the next call does not pay for them again): everything held by
 * engines and farm workers that are idle, and the calling thread's slice buffers.  Calls running on
 * other threads are not disturbed.  For hosts that embed the library and call it rarely (an HDF5
 * filter in a long-running service); never needed for correctness. */
void sperrhip_release(void);

/* Library/engine identification, e.g. "sperr_hip 0.1 (gfx950)". */
const char* sperrhip_version(void);

#ifdef __cplusplus
}
#endif
#endif
