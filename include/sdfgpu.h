/*
 * sdfgpu.h -- C ABI of the MI355X-native signed-distance-field build path.
 *
 * This is the drop-in boundary for the one hot path of UM-ARM-Lab/sdf_tools:
 *
 *   sdf_generation::ExtractSignedDistanceField      include/sdf_tools/sdf_generation.hpp:209-271
 *     = classify (:219-240) + BuildDistanceField x2 (:95-207, called :242-243)
 *       + signed merge / extrema (:245-269)
 *   its virtual-border variant                      include/sdf_tools/sdf_generation.hpp:273-420
 *   the CollisionMapGrid predicate                  include/sdf_tools/collision_map.hpp:680-712
 *
 * A maintainer binds these entry points from the reference's C++ (see
 * INTEGRATION.md); the in-tree mirror of the reference's C++ API
 * (include/sdf_tools/ headers) and the pysdf_tools module call nothing else.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary
 *   - voxel layout: index = x*ny*nz + y*nz + z (z fastest), the reference's
 *     VoxelGrid layout (src/sdf_tools/sdf.cpp:241-245, utils_3d.py:71-73)
 *   - every function returns SDFGPU_OK (0) or a negative sdfgpu_status;
 *     nothing throws; sdfgpu_last_error() gives the message for the handle
 *   - a handle is bound to one GPU; use one handle per host thread
 *   - builds on one handle share its scratch fields and status block.  Builds issued on the same stream are
 *     ordered by the stream; a build issued on a different stream than the previous one first waits (on the
 *     device, hipStreamWaitEvent) for the previous build's last kernel.  The tiered stage entry points
 *     (sdfgpu_sweep_zy_device / sdfgpu_sweep_zy_tiered_device, sdfgpu_sweep_x_lines_device) use the status block and
 *     scratch fields too and take part in the same ordering (they wait for the handle's previous work when that ran
 *     on another stream, and later work waits for them).  The remaining stage-level entry points
 *     (sdfgpu_sweep_x_device, sdfgpu_dense_ball_device, sdfgpu_slab_dense_phase ...) write only caller-owned
 *     buffers plus the handle's extrema slots: issue all stage calls of one build on one stream.
 *   - host-buffer entry points use the caller's output buffer as scratch while they run (its pages are
 *     faulted in while the input travels to the GPU): when such a call fails, the contents of out_sdf /
 *     out_grad are undefined
 *   - there is NO CPU fallback: without a usable HIP device sdfgpu_create fails
 */
#ifndef SDFGPU_H
#define SDFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdfgpu_context* sdfgpu_handle;

typedef enum sdfgpu_status {
    SDFGPU_OK = 0,
    SDFGPU_ERR_INVALID_ARGUMENT = -1, /* null pointer, non-positive dims, bad stride ...          */
    SDFGPU_ERR_HIP = -2,              /* a HIP runtime call failed (maps to std::runtime_error)   */
    SDFGPU_ERR_UNSUPPORTED_SIZE = -3, /* a dim > 16384 or nx^2+ny^2+nz^2 >= 2^30                   */
    SDFGPU_ERR_NO_DEVICE = -4,        /* no HIP device / device index out of range                */
    SDFGPU_ERR_UNRESOLVED = -5,       /* slab x-sweep needed rows beyond the supplied halo        */
    SDFGPU_ERR_REDZONE = -6           /* red-zone mode: a kernel of this call stored outside a buffer of the library (the message names it) */
} sdfgpu_status;

/* Library version, e.g. "sdfgpu 0.1 (gfx950)". */
const char* sdfgpu_version(void);

/* Number of visible HIP devices (0 if none / runtime unusable). */
int sdfgpu_device_count(void);

/* Create / destroy a context on `device`.  The context owns the device scratch
 * (int16 z-sweep field, int32 yz-sweep field, staging buffers) and re-uses it
 * across calls, growing on demand -- needed for the 30 Hz streaming use. */
int sdfgpu_create(int device, sdfgpu_handle* out_handle);
int sdfgpu_destroy(sdfgpu_handle h);

/* Message for the last non-OK status returned on this handle (never NULL).
 * With h == NULL returns the message of the last failed sdfgpu_create. */
const char* sdfgpu_last_error(sdfgpu_handle h);

/* ---------------------------------------------------------------------------
 * Whole-path entry points, host buffers.
 * Replaces sdf_generation::ExtractSignedDistanceField (sdf_generation.hpp:273-420)
 * once the caller has evaluated its predicate into `filled` (nonzero = filled),
 * in the reference's x->y->z order (:221-239).
 *   out_sdf : N floats, same layout.  sdf = filled ? -res*sqrt(D_free) : +res*sqrt(D_filled),
 *             computed as float(double) exactly like :254-265; +/-inf when a class is empty.
 *   out_max/out_min : extrema of the un-narrowed doubles (:246-269); with
 *             add_virtual_border the (free.max, filled.min) pair of :416-418.
 * ------------------------------------------------------------------------- */
int sdfgpu_build(sdfgpu_handle h, const uint8_t* filled,
                 int64_t nx, int64_t ny, int64_t nz,
                 double resolution, int add_virtual_border,
                 float* out_sdf, double* out_max, double* out_min);

/* Fast path for CollisionMapGrid::ExtractSignedDistanceField
 * (collision_map.hpp:680-712): `cells` is the grid's raw data
 * (GetImmutableRawData()), records of `cell_stride` bytes whose float occupancy
 * sits at `occupancy_offset`.  Classified on the device with exactly
 * occupancy > 0.5f || (unknown_is_filled && occupancy == 0.5f). */
int sdfgpu_build_cells(sdfgpu_handle h, const void* cells,
                       size_t cell_stride, size_t occupancy_offset, int unknown_is_filled,
                       int64_t nx, int64_t ny, int64_t nz,
                       double resolution, int add_virtual_border,
                       float* out_sdf, double* out_max, double* out_min);

/* Next-row N4: the predicates of TaggedObjectCollisionMapGrid (tagged_object_collision_map.hpp:730-856)
 * on raw TAGGED_OBJECT_COLLISION_CELL records {float occupancy; uint32 component; uint32 object_id;
 * uint32 convex_segment}.  A cell is filled iff its occupancy says so AND its object id passes:
 *   object_mode 0: any object                    (free_sdf_filled_fn :736-749)
 *   object_mode 1: object_id > 0                 (object_filled_fn :757-775, "named objects")
 *   object_mode 2: object_id in object_ids[0..n) (ExtractSignedDistanceField(objects_to_use) :817-827;
 *                                                 n == 0 means any object, like :826)
 * object_ids may hold any number of ids in any order (a sorted copy is searched on the device).
 * Classified on the device, then the same build as sdfgpu_build.
 * cells may be NULL: the records uploaded by the previous sdfgpu_build_tagged_cells call on this handle are used again
 * (same nx * ny * nz * cell_stride, no other host-buffer entry point on the handle in between; INVALID_ARGUMENT
 * otherwise).  A caller that builds one field per object (MakeObjectSDFs :875-891, one call per id) then sends the
 * 16 B/voxel records over PCIe once instead of once per object. */
int sdfgpu_build_tagged_cells(sdfgpu_handle h, const void* cells,
                              size_t cell_stride, size_t occupancy_offset, size_t object_id_offset,
                              int object_mode, const uint32_t* object_ids, int64_t n_object_ids,
                              int unknown_is_filled,
                              int64_t nx, int64_t ny, int64_t nz,
                              double resolution, int add_virtual_border,
                              float* out_sdf, double* out_max, double* out_min);

/* ---------------------------------------------------------------------------
 * Device-pointer variants (benchmark / streaming / multi-GPU callers).
 * All pointers are device pointers on the handle's GPU; `stream` is a
 * hipStream_t (NULL = default stream).  Asynchronous: kernels are enqueued on
 * `stream`, nothing is copied to the host.  Call sdfgpu_get_extrema afterwards
 * (it synchronises `stream`) to obtain (max, min) of the most recent build.
 * ------------------------------------------------------------------------- */
int sdfgpu_build_device(sdfgpu_handle h, const uint8_t* d_filled,
                        int64_t nx, int64_t ny, int64_t nz,
                        double resolution, int add_virtual_border,
                        float* d_out_sdf, void* stream);

int sdfgpu_build_cells_device(sdfgpu_handle h, const void* d_cells,
                              size_t cell_stride, size_t occupancy_offset, int unknown_is_filled,
                              int64_t nx, int64_t ny, int64_t nz,
                              double resolution, int add_virtual_border,
                              float* d_out_sdf, void* stream);

int sdfgpu_get_extrema(sdfgpu_handle h, double* out_max, double* out_min);

/* Bits in (round 6): the occupancy as ONE BIT per voxel in linear voxel order -- bit (v & 31) of 32-bit word (v >> 5) is set
 * iff voxel v = x*ny*nz + y*nz + z is filled; ceil(nx*ny*nz / 32) words, bits beyond the last voxel ignored.  This is what the
 * predicate loop of sdf_generation.hpp:219-240 produces when a caller keeps its occupancy packed (an octree leaf mask, a
 * voxel hash), what the host-buffer entry points above upload after classifying on the host, and what
 * sdfgpu_voxelize_points_bits_device writes.  The dense tier reads the field in place (where nz = 32 * 2^k the linear field IS
 * its [x][y][nz/32] bit field: no pack kernel runs; 16-byte alignment avoids one device copy), the z sweep and the generic dense
 * kernels read it through bit loaders: no byte mask exists anywhere.  Same results, extrema and errors as sdfgpu_build_device /
 * sdfgpu_build.  d_bits / bits must be 4-byte aligned. */
int sdfgpu_build_bits_device(sdfgpu_handle h, const uint32_t* d_bits,
                             int64_t nx, int64_t ny, int64_t nz,
                             double resolution, int add_virtual_border,
                             float* d_out_sdf, void* stream);
int sdfgpu_build_bits(sdfgpu_handle h, const uint32_t* bits,
                      int64_t nx, int64_t ny, int64_t nz,
                      double resolution, int add_virtual_border,
                      float* out_sdf, double* out_max, double* out_min);

/* Pageable host memory <-> device memory at the rate of the link.  These are the copies the host-buffer entry points
 * above use; wrappers that keep their own device buffers (the multi-GPU library, a caller filling a std::vector such
 * as the reference's GetImmutableRawData() storage, sdf.hpp / voxel_grid.hpp:760) can use them too.
 *   to_host:   the destination may be memory nobody has touched yet (a fresh std::vector / numpy array): a plain
 *              hipMemcpy then takes one first-touch page fault after the other (512^3 floats: ~40 ms); here the DMA lands
 *              in pinned staging chunks of the handle and a team of host threads copies them out while the next chunk
 *              is in flight (512^3 floats: ~11 ms = the PCIe transfer).
 *   from_host: a synchronous hipMemcpy from pageable memory is staged by one host thread (~5 GB/s measured); here a
 *              team fills the pinned chunks (128 MiB: 3 ms instead of 24 - 30).
 * Both are enqueued on `stream` behind the work already there and return when the transfer is complete.  One transfer
 * at a time per handle (they share the handle's staging chunks). */
int sdfgpu_copy_to_host(sdfgpu_handle h, void* dst, const void* d_src, size_t bytes, void* stream);
int sdfgpu_copy_from_host(sdfgpu_handle h, void* d_dst, const void* src, size_t bytes, void* stream);

/* Host occupancy -> device byte mask, classified ON THE HOST while the pinned staging chunks are filled (round 5): the team
 * of host threads evaluates the predicate -- mask byte != 0 (filled != NULL), or the CollisionMapGrid predicate
 * occupancy > 0.5f || (unknown_is_filled && occupancy == 0.5f) on raw cell records (cells != NULL; reference
 * include/sdf_tools/collision_map.hpp:689-704) -- into ONE BIT per voxel, 1/8 byte per voxel crosses PCIe (16 MiB for 512^3
 * instead of 128 MiB of mask or 1 GiB of 8-byte cells), and a kernel on `stream` spreads the bits into d_mask[n] (0 / 1).
 * This is what sdfgpu_build / sdfgpu_build_cells / *_to_device do with their input (option "host_pack" = 0 restores the
 * upload-and-classify-on-device path); exported for wrappers with their own device buffers (libsdfgpu_multi's per-rank
 * slabs).  Exactly one of filled / cells is non-NULL.  Returns when the host buffer has been consumed. */
int sdfgpu_upload_classified(sdfgpu_handle h, const uint8_t* filled, const void* cells, size_t cell_stride, size_t occupancy_offset,
                             int unknown_is_filled, int64_t n_voxels, uint8_t* d_mask, void* stream);

/* ---------------------------------------------------------------------------
 * Stage-level entry points for the x-slab multi-GPU path.
 * The grid is partitioned along x (the slowest axis); a rank owns rows
 * [x0, x0+nxs).  The z and y sweeps are slab-local:
 *
 *   sdfgpu_sweep_zy_device : mask slab [nxs,ny,nz] -> signed squared in-plane
 *       distance, int32 [nxs,ny,nz]  (+D for free voxels, -D for filled,
 *       magnitude SDFGPU_DSQ_INF where the plane holds no opposite voxel).
 *
 * The caller exchanges `halo` boundary planes of that field with its x
 * neighbours (RCCL send/recv) into one contiguous buffer
 * [halo_lo + nxs + halo_hi, ny, nz] and runs
 *
 *   sdfgpu_sweep_x_device  : x sweep + signed merge over the extended buffer,
 *       writing the nxs owned rows.  `lo_truncated` / `hi_truncated` say that
 *       real grid rows exist beyond the buffer on that side.  A voxel whose
 *       search would have needed such a row raises bit 0 of *d_status (uint32,
 *       device, caller-zeroed); the caller all-reduces it and, if set, widens
 *       the halo (or gathers whole lines) and re-runs.  x_global is the grid x
 *       of the first owned row and nx_global the full grid extent (virtual
 *       border clamp needs both).  Extrema of the owned rows go to
 *       d_maxdsq[2] (uint32 device: max d^2 over free, over filled voxels;
 *       caller-zeroed; all-reduce MAX then sdfgpu_extrema_from_dsq).
 * ------------------------------------------------------------------------- */
#define SDFGPU_DSQ_INF (1 << 30)

int sdfgpu_sweep_zy_device(sdfgpu_handle h, const uint8_t* d_filled,
                           int64_t nxs, int64_t ny, int64_t nz,
                           int32_t* d_plane_dsq, void* stream);

/* Same as sdfgpu_sweep_zy_device, and reports which y sweep ran: *d_far (uint32 device word, may be NULL) is set to 1
 * when the device-side probe found the slab far-field and the envelope kernel did the y sweep (a hint that the x
 * sweep will need whole lines too), else to 0.  sdfgpu_sweep_zy_device is this call with d_far = NULL. */
int sdfgpu_sweep_zy_tiered_device(sdfgpu_handle h, const uint8_t* d_filled,
                                  int64_t nxs, int64_t ny, int64_t nz,
                                  int32_t* d_plane_dsq, uint32_t* d_far, void* stream);

/* Exact x sweep + signed merge on COMPLETE lines of a y slab: d_plane_dsq is [nx][nys][nz] (what an all-to-all
 * re-partition of the x-slab plane fields delivers), rows y_global .. y_global + nys of a grid
 * whose y extent is ny_global (the virtual border needs both).  Any distance: the marching sweep (bounded scan)
 * or the envelope kernel, chosen on the device.  d_out_sdf: [nx][nys][nz]; d_maxdsq[2] as in
 * sdfgpu_sweep_x_device. */
int sdfgpu_sweep_x_lines_device(sdfgpu_handle h, const int32_t* d_plane_dsq,
                                int64_t nx, int64_t nys, int64_t nz,
                                int64_t y_global, int64_t ny_global,
                                double resolution, int add_virtual_border,
                                float* d_out_sdf, uint32_t* d_maxdsq, void* stream);

int sdfgpu_sweep_x_device(sdfgpu_handle h, const int32_t* d_plane_dsq,
                          int64_t halo_lo, int64_t nxs, int64_t halo_hi,
                          int64_t ny, int64_t nz,
                          int lo_truncated, int hi_truncated,
                          int64_t x_global, int64_t nx_global,
                          double resolution, int add_virtual_border,
                          float* d_out_sdf, uint32_t* d_maxdsq, uint32_t* d_status,
                          void* stream);

/* Dense-scene stages (see sdf_tools_amd/csrc/sdfgpu_dense.hpp), also usable on x slabs:
 *   sdfgpu_pack_bits_device : n_rows z-rows of nz occupancy bytes -> n_rows * nz / 32 words (bit i of
 *       word w = voxel z = 32 w + i is filled); nz % 32 == 0.
 *   sdfgpu_dense_ball_device: bit planes [rows_x, ny, nz/32] -> fp32 SDF of planes [out_lo, out_hi),
 *       exact for every voxel whose nearest opposite-class voxel is within squared distance 8
 *       (needs 2 planes of context on each side, i.e. halo planes from the x neighbours in slab
 *       mode; at a true grid face the buffer simply ends).  Raises *d_uncertified (uint32, caller-
 *       zeroed) if some voxel is farther than that -- the caller must then run the general path.
 *       d_maxdsq[2] as in sdfgpu_sweep_x_device.  nz must be 32 * 2^k <= 2048. */
int sdfgpu_pack_bits_device(sdfgpu_handle h, const uint8_t* d_filled, int64_t n_rows, int64_t nz,
                            uint32_t* d_bits, void* stream);
int sdfgpu_dense_ball_device(sdfgpu_handle h, const uint32_t* d_bits, int64_t rows_x, int64_t out_lo,
                             int64_t out_hi, int64_t ny, int64_t nz, double resolution,
                             float* d_out_sdf, uint32_t* d_maxdsq, uint32_t* d_uncertified, void* stream);
/* The kernels behind sdfgpu_sweep_x_device / sdfgpu_dense_ball_device collect their maxima in a slot array
 * (see sdfgpu_kernels.hpp: slot_max2) and each call ends with a one-block fold into d_maxdsq[2].  A caller
 * that issues several such calls per build (interior + border planes of a slab) can set the option
 * "defer_fold" = 1 and fold once itself: */
int sdfgpu_fold_extrema_device(sdfgpu_handle h, uint32_t* d_maxdsq, void* stream);

/* One x slab of the dense path in three calls (the same kernels as the stage entry points above, fewer host
 * round trips per build).  d_bits_ext: [halo_lo + nxs + halo_hi][ny][nz/32] with halo_lo / halo_hi = 0 or 2 planes
 * the caller fills by exchanging boundary planes; d_small: 4 words {max d^2 free, max d^2 filled, -, uncertified}.
 *   phase 0   clear d_small, pack the boundary planes (everything if the slab has no neighbour or is too thin)
 *             -> the caller posts the halo exchange of the 2 + 2 boundary bit-planes
 *   phase 1   pack the interior, ball kernel on the planes that need no neighbour data (10 / 11: only the first /
 *             second of the two)
 *   phase 2   (after the exchange has completed) ball kernel on the border planes, fold of the maxima */
int sdfgpu_slab_dense_phase(sdfgpu_handle h, int phase, const uint8_t* d_mask_slab,
                            int64_t nxs, int64_t ny, int64_t nz,
                            uint32_t* d_bits_ext, int64_t halo_lo, int64_t halo_hi,
                            double resolution, float* d_out_sdf, uint32_t* d_small, void* stream);

/* (max, min) from the two integer maxima (0 = class absent, >= SDFGPU_DSQ_INF =
 * infinite), reproducing sdf_generation.hpp:246-269 / :416-418. */
int sdfgpu_extrema_from_dsq(uint32_t max_dsq_free, uint32_t max_dsq_filled,
                            double resolution, double* out_max, double* out_min);

/* ---------------------------------------------------------------------------
 * Next-row N1: grid-aligned gradient of a device-resident field,
 * SignedDistanceField::GetGridAlignedGradient (include/sdf_tools/sdf.hpp:432-526)
 * for every voxel at once.  d_out_grad: [nx,ny,nz,3] float64 when
 * out_is_f64 != 0 (bit-identical to the reference's doubles) else float32.
 * Interior voxels use central differences; edge voxels use the clamped
 * one-sided form when enable_edge_gradients != 0, else are written as NaN
 * (the reference returns an empty vector there).
 * ------------------------------------------------------------------------- */
int sdfgpu_gradient_device(sdfgpu_handle h, const float* d_sdf,
                           int64_t nx, int64_t ny, int64_t nz,
                           double resolution, int enable_edge_gradients,
                           void* d_out_grad, int out_is_f64, void* stream);

/* Host-buffer form (what SignedDistanceField::GetFullGradient's fast path and pysdf_tools call instead of
 * nx*ny*nz host GetGradient calls, reference sdf.hpp:341-358 / utils_3d.py:77-90): sdf = N floats,
 * out_grad = N x 3 doubles (out_is_f64) or floats; NaN where the reference returns an empty vector. */
int sdfgpu_gradient(sdfgpu_handle h, const float* sdf,
                    int64_t nx, int64_t ny, int64_t nz,
                    double resolution, int enable_edge_gradients,
                    void* out_grad, int out_is_f64);

/* Next-row N1, query side: batched SignedDistanceField::EstimateDistance4d (sdf.hpp:947-961: trilinear
 * inter/extrapolation :836-902 of the 8 surrounding cell centres, each shrunk by half a cell toward the
 * surface :773-796, neighbour pairs per axis :798-833) and GetGradient4d (:383-430) at n world-frame
 * points (d_points: n x 3 doubles).  world_to_grid: 12 host doubles, row-major 3x4 = inverse origin
 * transform (NULL = identity); grid_to_world_rotation: 9 host doubles, row-major (NULL = identity).
 * Outputs (device, any may be NULL):
 *   d_distance[n]  the estimate, or oob_value where the point is outside the grid
 *   d_gradient[3n] world-frame gradient of the cell holding the point; NaN where the reference returns
 *                  an empty vector (outside, or boundary shell without enable_edge_gradients)
 *   d_flags[n]     bit0 = point inside the grid, bit1 = gradient available */
int sdfgpu_query_points_device(sdfgpu_handle h, const float* d_sdf,
                               int64_t nx, int64_t ny, int64_t nz, double resolution,
                               const double* world_to_grid, const double* grid_to_world_rotation,
                               float oob_value, const double* d_points, int64_t n_points,
                               int enable_edge_gradients,
                               double* d_distance, double* d_gradient, uint8_t* d_flags, void* stream);

/* Host-buffer form of the query above against a field that LIVES IN HBM (round 4; what the C++ mirror's
 * sdf_tools::DeviceSignedDistanceField::EstimateDistanceBatch / GetGradientBatch call, i.e. N1 for callers of the
 * reference's host-side SignedDistanceField::EstimateDistance* / GetGradient*, sdf.hpp:922-961, :383-430): points
 * (n x 3 doubles) and the three outputs are HOST arrays (any output may be NULL); d_sdf is a device pointer, e.g.
 * one filled by sdfgpu_build_device into memory from sdfgpu_device_malloc.  A caller that only needs answers at its
 * points never downloads the field (512 MiB at 512^3, ~10 ms of PCIe).  Synchronous.  Runs on the device's null stream behind this
 * handle's last build; a d_sdf written by ANOTHER producer on a non-blocking stream (another handle's build on a PyTorch stream, a
 * caller's kernel) must be complete before the call: synchronise that stream first. */
int sdfgpu_query_points(sdfgpu_handle h, const float* d_sdf,
                        int64_t nx, int64_t ny, int64_t nz, double resolution,
                        const double* world_to_grid, const double* grid_to_world_rotation, float oob_value,
                        const double* points, int64_t n_points, int enable_edge_gradients,
                        double* out_distance, double* out_gradient, uint8_t* out_flags);

/* Host input -> DEVICE-RESIDENT result (round 4): sdfgpu_build / sdfgpu_build_cells without the download of the field.
 * d_out_sdf is device memory for nx*ny*nz floats (sdfgpu_device_malloc); the extrema come back as in sdfgpu_build.
 * What sdf_generation::ExtractSignedDistanceFieldDevice and CollisionMapGrid::ExtractSignedDistanceFieldDevice call. */
int sdfgpu_build_to_device(sdfgpu_handle h, const uint8_t* filled,
                           int64_t nx, int64_t ny, int64_t nz, double resolution, int add_virtual_border,
                           float* d_out_sdf, double* out_max, double* out_min);
int sdfgpu_build_cells_to_device(sdfgpu_handle h, const void* cells, size_t cell_stride, size_t occupancy_offset,
                                 int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                 int add_virtual_border, float* d_out_sdf, double* out_max, double* out_min);

/* Device memory for callers without a HIP runtime of their own (the C++ mirror is plain host C++): memory on the
 * handle's device, usable with every *_device entry point and with sdfgpu_copy_to_host / sdfgpu_copy_from_host. */
int sdfgpu_device_malloc(sdfgpu_handle h, size_t bytes, void** out_ptr);
int sdfgpu_device_free(sdfgpu_handle h, void* ptr);

/* CollisionMapGrid predicate (collision_map.hpp:689-704) on raw cell records -> byte mask (1 = filled), for callers
 * of the slab stages, which take masks: occupancy > 0.5f || (unknown_is_filled && occupancy == 0.5f). */
int sdfgpu_classify_cells_device(sdfgpu_handle h, const void* d_cells, size_t cell_stride, size_t occupancy_offset,
                                 int unknown_is_filled, int64_t n_cells, uint8_t* d_mask, void* stream);

/* Next-row N2: point cloud -> occupancy grid, the convention of scripts/3d_sdf_demo_rviz.py:22-29:
 * index = trunc((p - origin) / resolution) per axis (fp64 arithmetic on fp32 points), mask[ix][iy][iz] = 1
 * with explicit x, y, z axis order; points outside the grid are dropped.  d_points: n_points x 3 floats
 * (x, y, z interleaved).  clear_first != 0 zeroes the mask before scattering.  Feeds
 * sdfgpu_build_device without leaving the GPU (the streaming configuration). */
int sdfgpu_voxelize_points_device(sdfgpu_handle h, const float* d_points, int64_t n_points,
                                  const double* origin, double resolution,
                                  int64_t nx, int64_t ny, int64_t nz,
                                  uint8_t* d_mask, int clear_first, void* stream);

/* The same scatter into the bit field of sdfgpu_build_bits_device (one atomic OR per point; clear_first zeroes
 * ceil(nx*ny*nz / 32) words): a streaming frame goes point cloud -> bits -> SDF without a byte mask in between. */
int sdfgpu_voxelize_points_bits_device(sdfgpu_handle h, const float* d_points, int64_t n_points,
                                       const double* origin, double resolution,
                                       int64_t nx, int64_t ny, int64_t nz,
                                       uint32_t* d_bits, int clear_first, void* stream);

/* Red zones (round 6).  With SDFGPU_REDZONE=1 in the environment when sdfgpu_create runs -- or after
 * sdfgpu_set_option(h, "redzone", 1) -- every device allocation of the library (scratch fields, status block, extrema slots,
 * staging buffers, sdfgpu_device_malloc memory) carries 4 KiB of canary bytes in front and behind, and every entry point that
 * may have launched a kernel ends with one check kernel over all of them and a synchronisation: a store outside a buffer
 * fails THAT call with SDFGPU_ERR_REDZONE and a message that names the buffer and the offset.  Debug mode (calls become
 * synchronous; results are unchanged).  sdfgpu_redzone_check runs the same check on demand (wrappers with device buffers from
 * sdfgpu_device_malloc: libsdfgpu_multi); SDFGPU_OK when the mode is off. */
int sdfgpu_redzone_check(sdfgpu_handle h, void* stream);

/* Debug / test hooks: copy the intermediates of the most recent
 * sdfgpu_build*_device call to host buffers (N int16 / N int32). */
int sdfgpu_debug_copy_zsweep(sdfgpu_handle h, int16_t* out_host, int64_t n);
/* ... and float(sqrt((double)D) * resolution) (sdf_generation.hpp:254-265) for D = 0 .. n - 1 (n <= 2^24) into d_out[n] (device):
 * fast = 0 the fp64 sequence of the far-field x sweep, fast = 1 the fp32 form of sdf_tools_amd/csrc/sdfgpu_finish.hpp (round 6: exact,
 * measured slower than the fp64 sequence on MI355X and therefore not used by the kernels; kept with its exhaustive tests);
 * *out_slow_lanes = lanes of the fp32 form that asked for the fp64 sequence. */
int sdfgpu_debug_finish_table(sdfgpu_handle h, float* d_out, int64_t n, double resolution, int fast, uint32_t* out_slow_lanes);
int sdfgpu_debug_copy_yzsweep(sdfgpu_handle h, int32_t* out_host, int64_t n);
/* The device-side habit of the far-field y sweep's two-valued tiles (option "flat_tiles" = 1; synchronises with the last build):
 * out_score = the votes so far -- clamped to [0, 256] in front of every y sweep that consults it, tiles that tried and did not qualify
 * add 1, tiles that qualified subtract 3 -- and out_gate = what the last such sweep was told (1: every candidate tile tries, 0: every
 * 64th).  Results never depend on either. */
int sdfgpu_debug_flat_habit(sdfgpu_handle h, int* out_score, int* out_gate);

/* Per-stage timing with HIP events recorded on the build's own stream (bench.py's roofline leg).
 * While enabled, every sdfgpu_build*_device call brackets its seven stages with events:
 * [0] K0 pack, [1] KD dense ball kernel, [2] K1 z sweep, [3] K2 / K12 y (z+y) sweep, [4] KE2 envelope y
 * sweep, [5] K3 x sweep, [6] KE3 envelope x sweep (a stage that is not launched, or exits on its guard
 * flag, shows ~0).  sdfgpu_get_stage_times
 * synchronises, adds the elapsed times since the last call into out_ms_sum[7] (milliseconds),
 * returns the number of builds they cover in *out_builds and resets the accumulators.
 * enable = 2 brackets only the dominant kernel of the dense path (stage [1]; the other stages then read 0):
 * two events per build instead of five, for timed benchmark loops; enable = 3 does that on every 4th build
 * only (out_builds then counts the sampled builds). */
int sdfgpu_set_profiling(sdfgpu_handle h, int enable);
int sdfgpu_get_stage_times(sdfgpu_handle h, double* out_ms_sum, int64_t* out_builds);

/* Named integer options.  EVERY option leaves the results exact: they move work between kernels, switch a measured optimisation
 * off for an A/B, or put the handle's policy into a state a test needs.  Unknown names return SDFGPU_ERR_INVALID_ARGUMENT.
 * Switches that SKIP work for profiling ("dc_debug", "dc_debug_stage", "ball_variant") exist only in libraries built with
 * -DSDFGPU_DEBUG_HOOKS (tools/probe/libsdfgpu_hooks.so) and are rejected by the shipped one.
 * [T] = test / fuzz only (forces a state the policy reaches by itself), [AB] = A/B switch of a measured optimisation (default = the
 * faster setting; DESIGN.md / LAB_NOTES.md hold the measurement), [U] = for users.
 *
 *  tier selection                     default  meaning
 *  "dense"                   [U]      1        try the bit-parallel dense tier first (0: sweeps / far-field pair only)
 *  "dense_generic"           [AB]     1        shapes the tuned dense kernels do not take (nz not 32 * 2^k) use their generic forms
 *  "dense_retry"             [U]      16       after an uncertified dense attempt, try again only every N-th build (0: always)
 *  "envelope"                [AB]     1        bound the marching scans and redo far-field sweeps with k_envelope_dc (0: unbounded scans)
 *  "envelope_dc"             [AB]     1        0: never use the far-field kernel
 *  "envelope_mode"           [T]      0        1: the far-field kernel is the only sweep of both axes, no probes
 *  "far_predict"             [U]      1        handles whose recent builds were far-field skip probes + marching launches (0 off, 2 always)
 *  "far_threshold_y/_x", "far_fraction_den_y/_x"  [AB]  16, 9 / 5, 24   an axis is far-field when > 1/den of the probed voxels have d^2 >= thr
 *  "mid_threshold_y", "mid_fraction_den_y"        [AB]  16 / 24         radius-8 y window when > 1/den of them have d^2 >= thr (den 0: never)
 *  "probe_window"            [AB]     1        tier probes as window statistics (0: level A of the far-field search on sampled tiles)
 *  "policy_reset"            [T]      -        forget what the handle learned from earlier builds
 *  "expect_dense"            [T]      0        put the handle into the "dense tier trusted" state (stand-by pair behind it)
 *  "fixup_mode", "dense3_mode" [T]    0        force the fix-up stage / KD3 in KD's place with the next build
 *
 *  dense tier
 *  "fixup"                   [AB]     1        fix-up kernel KF behind the ball kernel for almost-dense scenes
 *  "dense3"                  [AB]     1        wide ball kernel KD3 (|offset| <= 3) as the fix-up stage's first kernel
 *  "dense3_staged"           [AB]     1        builds that cannot expect KD to decide the scene carry KD3 + KF behind it, guarded
 *  "dense3_fixed"            [AB]     1        KD3's nz = 512 instance (compile-time row pitch)
 *  "dense_shell"             [AB]     1        shell pass KD6 (16 <= d^2 <= 36) between KD3 and KF
 *  "shell_min_words", "shell_budget_den"  [AB]  128 / 8   KD6: open words below which a tile group is left to KF; budget 1/den of the voxels
 *  "standby_far"             [AB]     1        stand-by behind a trusted dense tier = far-field pair (0: fused z+y + marching x, unbounded)
 *  "standby_fold"            [AB]     1        the stand-by x sweep's launch also folds the extrema (one launch less per build)
 *  "standby_grid"            [AB]     1024     workgroups of the stand-by launches
 *  "pack_variant", "ball_block", "nt_store"  [AB]  0   K0 unroll / KD workgroup size / non-temporal output stores
 *
 *  sweeps
 *  "fused_zy"                [AB]     1        let the policy use the fused z+y kernel (0 never, 2 always when the shape allows)
 *  "fused_window"            [AB]     2        its register-window radius at nz = 512 (2 or 3)
 *  "plane16"                 [AB]     1        int16 plane field + int32 side table between the y and x sweeps (0: int32 plane field)
 *  "y16"                     [AB]     1        y sweep of that pipeline through the packed 16-bit kernel
 *  "z_wave"                  [AB]     1        z sweep with whole rows per wave where nz = 64 ... 1024
 *  "x16_voxels_per_lane", "x16_window", "march_window"  [AB]  4 / 3 / 3   K3/16 variant; forced radius-8 windows (= 8)
 *  "rows_per_chunk_y/_x/_zy" [AB]     0        rows marched per thread (0: automatic); also sdfgpu_set_tuning
 *  "i32_handoff"             [AB]     1        far-field pair hands exact int32 plane values from the y to the x sweep
 *  "dc_fixed"                [AB]     1        far-field kernel: instances with the 512- / 1024-voxel line geometry at compile time
 *  "plane_skip"              [AB]     1        builds that go straight to the far-field pair skip the x-planes without a filled voxel
 *  "flat_tiles"              [AB]     1        ... and their y sweep skips its search in tiles whose lines hold at most two values outside
 *                                              their zero sites (a floor under open space, table tops, boxes in open space); tried on
 *                                              grids with a floor and, for y lines longer than 512, on any scene; 1: while it pays (a
 *                                              device-side habit, sdfgpu_debug_flat_habit), 2: every candidate tile tries, 0: never
 *
 *  host side / debugging
 *  "host_pack"               [U]      1        host-buffer builds classify on the host and upload 1 bit / voxel (0: upload + classify on
 *                                              the device; 2: whatever the size)
 *  "defer_fold"              [U]      0        stage entry points leave their maxima in the slots until sdfgpu_fold_extrema_device
 *  "redzone"                 [U]      0        canaries around every device allocation, checked at the end of every call (see above) */
int sdfgpu_set_option(sdfgpu_handle h, const char* name, int value);

/* Which kernels the most recent sdfgpu_build*_device call used: bit 0 = fused z+y kernel (K12),
 * bit 1 = 16-bit plane field (K3/16), bit 2 = dense kernel (K0 + KD) enqueued in front, bit 3 = the guarded
 * stand-by behind a trusted dense tier was the far-field pair (K1 -> KE2 -> KE3, bounded on any scene), bit 4 = the
 * dense stage was its wide form (KD3 + fix-up kernel in KD's place), bit 5 = that form was enqueued BEHIND KD, guarded on
 * KD's verdict (a build that had no reason to expect that KD decides the scene), bit 6 = the far-field pair was enqueued
 * without probes and marching launches because the handle's recent builds were far-field on both axes (option
 * "far_predict": 0 never, 1 learnt -- the default --, 2 every build; exact either way, every 16th build probes again). */
int sdfgpu_last_build_info(sdfgpu_handle h, int* out_fused_zy);

/* Which path did the work of the last build (synchronises): bit 0 = the dense kernel decided every voxel
 * (the general pipeline behind it exited immediately); bit 1 / bit 2 = the y / x sweep was done by the far-field
 * kernel (chosen by the probe, after a marching sweep hit its scan bound, or as the stand-by pair); bits 8..13 = why the
 * dense tier handed the scene on (diagnostics): 8 a staged tile held one class only, 9 a wave without a single decided
 * voxel, 10 a wave with more undecided voxels than the fix-up kernel takes, 11 a tile over the fix-up kernel's cap,
 * 12 a voxel beyond the fix-up kernel's reach (d^2 > 64), 13 a voxel beyond the ball with no fix-up stage behind. */
int sdfgpu_last_dense_certified(sdfgpu_handle h, int* out_certified);

/* Tuning hook (benchmarks): rows marched per thread in the y / x sweeps
 * (0 = automatic). */
int sdfgpu_set_tuning(sdfgpu_handle h, int rows_per_chunk_y, int rows_per_chunk_x);

#ifdef __cplusplus
}
#endif
#endif /* SDFGPU_H */
