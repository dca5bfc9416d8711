/* include/uvol_codec.h — C ABI of the MI355X-native UVOL codec (libuvolcodec.so).
 *
 * This is the drop-in boundary for the ONE hot path of EtherealEngine/Universal-Volumetric:
 * the per-frame geometry + texture encode that scripts/Encoder.py performs by spawning two
 * external processes,
 *     draco_encoder -i f.obj -o f.drc -qp 11 -qt 10 -qn 8 -qg 8 -cl 7      (scripts/Encoder.py:260-262)
 *     basisu -ktx2 -tex_type video -multifile_printf P -multifile_num B -multifile_first i -y_flip
 *            -output_file texture_%07u.ktx2                                  (scripts/Encoder.py:290-292)
 * Every entry point below replaces one of those process boundaries with an in-process call that
 * runs hand-written HIP kernels on gfx950.  Conventions follow the reference's own in-repo C-ABI
 * precedent (deprecated/encoder_legacy/codec/corto_codec.h:41-43): opaque handle, caller-owned
 * buffers, int status, no exceptions, no torch types.
 *
 * Threading: a ctx is bound to one GPU and one HIP stream; it is NOT thread-safe.  Different
 * ctxs may be used from different threads / processes (one process per GPU in bench.py).
 * Every batched encode entry point also has an enqueue form (`*_async`): it returns at once, the work runs in call order on the
 * context, and uvol_sync(ctx) completes it (SURVEY 8b "Threading").
 * There is NO CPU fallback: uvol_ctx_create fails with UVOL_E_NODEVICE when no gfx950 GPU is present.
 */
#ifndef UVOL_CODEC_H
#define UVOL_CODEC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVOL_ABI_VERSION 1

enum {
  UVOL_OK = 0,
  UVOL_E_INVALID = -1,     /* bad argument */
  UVOL_E_NODEVICE = -2,    /* no usable HIP device (fails loudly, never falls back to CPU) */
  UVOL_E_HIP = -3,         /* HIP runtime error, see uvol_last_error */
  UVOL_E_NOSPACE = -4,     /* caller output buffer too small (out_len holds the required size) */
  UVOL_E_ENCODE = -5,      /* per-frame encode failure (bad indices, empty mesh, ...) */
  UVOL_E_UNSUPPORTED = -6
};

/* Mirrors the numeric project-config.json fields consumed on the hot path
 * (scripts/Encoder.py:171-179 template, :260 and :290 use sites; defaults identical). */
typedef struct uvol_params {
  int32_t q_position_attr;          /* Q_POSITION_ATTR,         default 11 */
  int32_t q_texture_attr;           /* Q_TEXTURE_ATTR,          default 10 */
  int32_t q_normal_attr;            /* Q_NORMAL_ATTR,           default 8  */
  int32_t q_generic_attr;           /* Q_GENERIC_ATTR,          default 8 (accepted, unused: no generic attribute on this ABI) */
  int32_t draco_compression_level;  /* DRACO_COMPRESSION_LEVEL 0..10, default 7.  0 = sequential connectivity + difference predictor (stock draco_encoder's choice at
                                       this level); 1..10 are encoded with the level-7 tool set (valence edgebreaker; the level itself is not part of the bitstream) */
  int32_t ktx2_batch_size;          /* KTX2_BATCH_SIZE (mandatory in the reference) = layers per .ktx2 */
  int32_t etc1s_quality;            /* basisu -q equivalent, 1..255, default 128 (Encoder.py passes none) */
  int32_t y_flip;                   /* basisu -y_flip (Encoder.py:290 always passes it), default 1 */
  int32_t max_batch;                /* frames in flight per geometry batch, default 32 */
  int32_t cu_mod;                   /* experimental CU partition (hipExtStreamCreateWithCUMask), 0 = all CUs (default).  > 1: mask bits with
                                       (index % cu_mod) in cu_residues; -1: the CUs with ordinal lo .. hi - 1 inside EVERY XCD, cu_residues =
                                       lo << 8 | hi.  Mask bit i is CU i / 8 of XCD i % 8, and a queue cannot be kept off an XCD
                                       (profiles/r05_xcd_census.json) */
  int32_t cu_residues;              /* bit r set = residue r allowed (cu_mod > 1); lo << 8 | hi (cu_mod == -1) */
  int32_t traverse_vbits_l2;        /* 1: the attribute traversers keep only their face bitmap in LDS and the vertex bitmap in L2
                                       (6 instead of 3 per CU): pays off when several contexts keep > 700 frames in flight */
  int32_t stream_priority;          /* 1: create the context's HIP stream with the highest priority (short, LDS-hungry texture
                                       kernels then get free CU slots before the long geometry walkers take them) */
  int32_t uastc;                    /* basisu -uastc: 1 = the texture entry points write UASTC LDR 4x4 .ktx2 files (DFD colour model 166, no
                                       supercompression; every layer an independent image) instead of ETC1S/BasisLZ; default 0 (Encoder.py:290 passes no -uastc) */
  int32_t reserved[2];
} uvol_params;

void uvol_params_default(uvol_params *p);

typedef struct uvol_ctx uvol_ctx;

int  uvol_abi_version(void);
int  uvol_device_count(void);
/* device = HIP ordinal. Returns UVOL_OK and *out on success. */
int  uvol_ctx_create(int device, const uvol_params *params, uvol_ctx **out);
void uvol_ctx_destroy(uvol_ctx *ctx);
const char *uvol_last_error(const uvol_ctx *ctx);
/* completes everything enqueued on ctx (uvol_*_async calls and the stream); returns the first error among the enqueued calls */
int  uvol_sync(uvol_ctx *ctx);

/* Page-locked host memory (SURVEY 8(d) times the path from "inputs resident in pinned host memory").  The entry points that take HOST arrays
 * copy pageable memory through the library's own pinned double buffers (host threads fill them, ~42 GB/s).  A call whose input arrays ALL lie
 * in memory from uvol_host_alloc takes the UPLINK instead (round 6): the uploads of all its groups (meshes) / parts (texture segments) are
 * queued on a copy stream of the context when the call begins, ahead of its kernels, and the encoders' streams wait for them on the device -
 * no host thread touches the data, and with the enqueue forms (`*_async`) the next call's uploads cross the link while this call encodes.
 * The device layout mirrors the caller's: arrays that lie back to back in host memory (256-byte aligned, gaps of <= 4 KiB) go over in ONE
 * copy per run - lay the arrays of a frame, and consecutive frames, next to each other in the arena.
 * One pageable array and the whole call is staged as before.  Input arrays belong to the call until it completes (blocking forms: until they
 * return; enqueue forms: until uvol_sync).  The ring of upload slots holds 8 mesh groups / 4 texture parts at most (a slot of 640
 * 100 k-vertex frames is 6.8 GB; uvol_trim gives them back).  Process-wide registry, thread-safe; NULL when the runtime refuses (no device,
 * out of lockable memory).  What a caller of the reference holds at this boundary are files / host buffers (scripts/Encoder.py:244-302):
 * this is where it would read them into. */
void *uvol_host_alloc(size_t bytes);
void  uvol_host_free(void *p);
/* uvol_sync, then the workspaces of ctx - geometry lanes, texture lanes, upload slots - go back to the device (they only grow: a context that
 * once ran a 2560-frame call as one group keeps ~130 GB until it is destroyed); its streams stay, the next call allocates what it needs.  No counterpart in the reference
 * (its encoders are processes that exit, scripts/Encoder.py:266-302); a long-lived host uses it between jobs of very different sizes. */
int  uvol_trim(uvol_ctx *ctx);

/* One frame of OBJ-shaped geometry: separate value arrays + per-corner index triplets
 * (what draco_encoder parses out of `v/vt/vn/f` lines).  uv / nrm (and their index arrays) may be
 * NULL.  Pointers are HOST pointers for uvol_encode_mesh* and DEVICE pointers for *_dev. */
typedef struct uvol_mesh {
  const float    *pos;      uint32_t n_pos;    /* n_pos * 3 */
  const float    *uv;       uint32_t n_uv;     /* n_uv  * 2 */
  const float    *nrm;      uint32_t n_nrm;    /* n_nrm * 3 */
  const uint32_t *idx_pos;                     /* 3 * n_faces */
  const uint32_t *idx_uv;                      /* 3 * n_faces or NULL */
  const uint32_t *idx_nrm;                     /* 3 * n_faces or NULL */
  uint32_t        n_faces;
} uvol_mesh;

/* Upper bound of the .drc size for a mesh (use to size `out`). */
size_t uvol_mesh_bound(const uvol_mesh *m);

/* Device memory one frame of these dimensions occupies while it is in flight inside uvol_encode_mesh_batch[_dev] (workspace with
 * lifetime-shared arrays + its share of the packed output area): frames in flight per GPU = HBM left / this. */
size_t uvol_mesh_workspace(const uvol_ctx *ctx, const uvol_mesh *m);

/* Replaces one `draco_encoder` process (scripts/Encoder.py:260-262): OBJ arrays -> .drc bytes. */
int uvol_encode_mesh(uvol_ctx *ctx, const uvol_mesh *mesh, uint8_t *out, size_t cap, size_t *out_len);

/* Batched form of HOT LOOP 1 (scripts/Encoder.py:256-267): n independent frames, all in flight on
 * the GPU at once (one serial connectivity walker per frame, parallel kernels batched over frames).
 * status[i] is the per-frame result; a failed frame does not poison the batch. */
int uvol_encode_mesh_batch(uvol_ctx *ctx, const uvol_mesh *meshes, int n,
                           uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status);
/* Same with inputs already resident in HBM (device pointers inside `meshes`; `meshes` itself is a host array). */
int uvol_encode_mesh_batch_dev(uvol_ctx *ctx, const uvol_mesh *meshes, int n,
                               uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status);

/* Ingest on the device (`draco_encoder -i frame.obj` parses the OBJ text itself, scripts/Encoder.py:256-262): n OBJ files
 * as TEXT in host memory -> meshes_out[i] with DEVICE pointers (v / vt / vn / f lines, polygons fanned, 1-based and negative indices;
 * bit-identical to the host parser of host/uvol_host.cpp, i.e. to strtof), ready for uvol_encode_mesh_batch_dev[_async].  The arrays
 * live in the context's slot `slot` (0 or 1) until that slot is parsed into again.  To parse batch b + 1 WHILE batch b encodes, parse on a
 * second context of the same device (contexts are independent and device pointers are not tied to one; host/uvolenc.cpp does this) -
 * on one context the call waits for that context's enqueued calls like every other entry point.  status[i]: UVOL_OK; UVOL_E_UNSUPPORTED = the text holds a number or line the device
 * parser leaves to the host (more than 19 significant digits, inf / nan, a value on a float rounding boundary, an incomplete `v` line):
 * parse that file with the host parser; UVOL_E_INVALID = a face references a missing vertex / no faces.  Blocking. */
int uvol_parse_obj_batch_dev(uvol_ctx *ctx, const uint8_t *const *obj_text, const size_t *lens, int n, int slot,
                             uvol_mesh *meshes_out, int *status);

/* The PNG half of the ingest stage (`basisu` reads the PNGs itself, scripts/Encoder.py:274-292): n images of one size as INFLATED
 * scanlines in host memory - what zlib gives for the concatenated IDAT chunks of an 8-bit, non-interlaced RGB (channels 3) or RGBA (4)
 * PNG: height rows of one filter-type byte + width * channels filtered bytes - are un-filtered on the device (Sub / Up / Average /
 * Paeth, bit-identical to the host reader) into RGBA8 layers, top row first, in the context's slot `slot` (0 / 1; valid until that slot
 * is used again): rgba_dev_out[i] is what uvol_encode_texture_segments_dev takes.  The inflate stays with the caller (one serial bit
 * stream per file; the files of a batch inflate in parallel on host threads).  Other PNG variants (16-bit, palette, grey, interlaced,
 * wider than 8192): decode them on the host.  The call returns once the kernel is queued on an ingest stream of the context (the host
 * buffers may be re-used at once: pageable buffers have been staged, buffers in uvol_host_alloc memory have been read - the call waits for
 * those copies, not for the kernels); the context's texture entry points order themselves behind it, a caller that
 * reads the layers itself calls uvol_sync(ctx) first.  So the un-filter of batch k + 1 overlaps the encode of batch k. */
int uvol_unfilter_png_batch_dev(uvol_ctx *ctx, const uint8_t *const *inflated, int n, uint32_t width, uint32_t height, int channels,
                                int slot, const uint8_t **rgba_dev_out);

/* The same with the zlib INFLATE on the device too (what `basisu` does first with every PNG it reads, scripts/Encoder.py:274-292): n images
 * of one size, each as its zlib stream - the concatenated data of the file's IDAT chunks, zlib header and Adler-32 trailer included -
 * in host memory.  One wave per stream inflates it (stored, fixed and dynamic blocks; the 32 KiB window lives in LDS) into the scanline
 * buffer the un-filter kernel then reads, so the host uploads the ~3 MB file instead of inflating it and staging 16.8 MB.  Same slots,
 * ordering and output as uvol_unfilter_png_batch_dev.  A corrupt stream, or one that does not hold height x (1 + width x channels) bytes,
 * or whose Adler-32 trailer does not match, fails ALONE: uvol_png_status reports it, its layer is undefined, the other images are not
 * affected. */
int uvol_inflate_png_batch_dev(uvol_ctx *ctx, const uint8_t *const *zlib_streams, const size_t *lens, int n, uint32_t width, uint32_t height,
                               int channels, int slot, const uint8_t **rgba_dev_out);
/* Per-image statuses of the last uvol_unfilter_png_batch_dev / uvol_inflate_png_batch_dev call on `slot` (waits for its kernels):
 * status[i] = UVOL_OK or UVOL_E_INVALID, n <= the images of that call.  status == NULL: the first failure is the return value. */
int uvol_png_status(uvol_ctx *ctx, int slot, int *status, int n);

/* GPU-resident form (SURVEY 8(b) "variants taking arrays of frames + hipStream_t"; caller-owned buffers as in
 * deprecated/encoder_legacy/codec/corto_codec.h:41-43): inputs are device pointers PRODUCED ON `producer_stream` (a hipStream_t passed
 * as void *, NULL = already complete) - the codec's kernels are ordered after the work queued on that stream so far, without a host
 * wait -, and the .drc bitstreams are LEFT IN HBM: packed into the caller's device buffer `dev_out` (capacity dev_cap bytes; 32768 +
 * 8 bytes per face per frame always suffices for the default quantisation), frame i at dev_out + out_offs[i], out_lens[i] bytes long;
 * offsets, lengths and status come back to the host, the payload never does.  A frame that does not fit gets UVOL_E_NOSPACE.  Blocking. */
int uvol_encode_mesh_batch_dev_out(uvol_ctx *ctx, const uvol_mesh *meshes, int n, void *producer_stream,
                                   uint8_t *dev_out, size_t dev_cap, size_t *out_offs, size_t *out_lens, int *status);

/* Enqueue forms.  The call returns as soon as its arguments are recorded (the `meshes`, `outs` and `caps` ARRAYS are copied);
 * the frames' input arrays, the output buffers and `out_lens` / `status` belong to the call until uvol_sync(ctx) returns, which
 * reports the first failing call (its message through uvol_last_error).  Calls enqueued on one ctx run in order; a blocking entry
 * point on the same ctx waits for them first.  A host driver overlaps its own work - parsing the next batch, writing the previous
 * one - with the GPU this way without owning a thread per context (host/uvolenc.cpp does). */
int uvol_encode_mesh_batch_async(uvol_ctx *ctx, const uvol_mesh *meshes, int n,
                                 uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status);
int uvol_encode_mesh_batch_dev_async(uvol_ctx *ctx, const uvol_mesh *meshes, int n,
                                     uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status);

/* Replaces one `basisu -ktx2 -tex_type video` process (scripts/Encoder.py:290-292):
 * n_layers RGBA8 images (width*height*4 bytes each, top row first as a PNG decoder yields them)
 * -> one ETC1S/BasisLZ .ktx2 with n_layers array layers (layer 0 I-frame, others P-frames).
 * A segment any of whose images has alpha != 255 gets ALPHA SLICES, as basisu writes them and the stock player reads them
 * (src/lib/KTX2Loader.js:493-497): a second slice per image (the alpha channel as a grey image through the same codebooks), a second
 * DFD sample (channel 15) and the image descs' second offset / length pair; opaque segments are unchanged, and the two kinds may
 * share a batch.  Read back by uvol_decode_texture_segments (RGBA32); the opaque targets (ETC1 / BC7) refuse such a file. */
size_t uvol_texture_bound(uint32_t width, uint32_t height, int n_layers);
int uvol_encode_texture_segment(uvol_ctx *ctx, const uint8_t *const *rgba, int n_layers,
                                uint32_t width, uint32_t height,
                                uint8_t *out, size_t cap, size_t *out_len);
/* Same with the layers already resident in HBM (array of device pointers). */
int uvol_encode_texture_segment_dev(uvol_ctx *ctx, const uint8_t *const *rgba_dev, int n_layers,
                                    uint32_t width, uint32_t height,
                                    uint8_t *out, size_t cap, size_t *out_len);

/* Batched form of HOT LOOP 2 (scripts/Encoder.py:279-298): n_segments independent segments of n_layers layers each
 * (rgba[s * n_layers + l]), all of one size, encoded with ONE kernel launch per pipeline stage. */
int uvol_encode_texture_segments(uvol_ctx *ctx, const uint8_t *const *rgba, int n_segments, int n_layers,
                                 uint32_t width, uint32_t height,
                                 uint8_t *const *outs, const size_t *caps, size_t *out_lens);
int uvol_encode_texture_segments_dev(uvol_ctx *ctx, const uint8_t *const *rgba_dev, int n_segments, int n_layers,
                                     uint32_t width, uint32_t height,
                                     uint8_t *const *outs, const size_t *caps, size_t *out_lens);

/* enqueue forms of the two batched texture entry points (same contract as uvol_encode_mesh_batch_async) */
int uvol_encode_texture_segments_async(uvol_ctx *ctx, const uint8_t *const *rgba, int n_segments, int n_layers,
                                       uint32_t width, uint32_t height,
                                       uint8_t *const *outs, const size_t *caps, size_t *out_lens);
int uvol_encode_texture_segments_dev_async(uvol_ctx *ctx, const uint8_t *const *rgba_dev, int n_segments, int n_layers,
                                           uint32_t width, uint32_t height,
                                           uint8_t *const *outs, const size_t *caps, size_t *out_lens);

/* ---- decode path, texture half ----
 * Replaces, for the RGBA32 target, what the stock player does per .ktx2 segment: KTX2Loader parses the container and
 * hands every array layer to the basis transcoder (reference src/lib/KTX2Loader.js:469-580; src/V2/player.ts:338-356
 * uploads the layers as one sampler2DArray).  Input: BasisLZ/ETC1S .ktx2 files as uvol_encode_texture_segment[s] or
 * `basisu -ktx2 -tex_type video` write them (with or without alpha slices, one mip level), or the UASTC .ktx2 files this codec writes with
 * uvol_params.uastc (told apart by the DFD colour model).  Output: RGBA8, rows in stored order. */
/* host-only: container dimensions of one file (UVOL_E_INVALID if it is not a KTX2/BasisLZ file this decoder handles;
 * a UASTC file whose level data are Zstandard-supercompressed - supercompressionScheme 2, the default of stock `basisu -uastc -ktx2` - is
 * read since round 5 when the system's libzstd.so.1 is installed: the decode / transcode entry points inflate the level on the host
 * (dlopen; this library contains no Zstandard code) and go on as for a scheme-0 file; without libzstd, for another scheme or a corrupt
 * frame the file is UVOL_E_UNSUPPORTED, as every supercompressed UASTC file was before) */
int uvol_ktx2_info(const uint8_t *ktx2, size_t len, uint32_t *width, uint32_t *height, uint32_t *layers);
/* n_segments files of one width / height / layer count; rgba[s * layers + l] receives width*height*4 bytes
 * (layer_cap = size of each buffer).  One kernel launch per stage for the whole batch. */
int uvol_decode_texture_segments(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments,
                                 uint8_t *const *rgba, size_t layer_cap);
/* Same, writing straight into device buffers (rgba_dev = host array of device pointers). */
int uvol_decode_texture_segments_dev(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments,
                                     uint8_t *const *rgba_dev, size_t layer_cap);

/* ETC1 target (the player's `etc2` raw-texture family, reference src/Interfaces.ts:19, src/V2/player.ts:338-356; any ETC2 sampler
 * reads ETC1 blocks): blocks[s * layers + l] receives ceil(w/4) * ceil(h/4) 8-byte ETC1 blocks in raster order (layer_cap =
 * size of each buffer); an exact re-pack, every ETC1S block is a valid ETC1 block.  outputs_on_device: blocks are device pointers.
 * UASTC sources (round 5; the loader's etc2Supported / etc1Supported rows for UASTC, src/lib/KTX2Loader.js:619-636): this entry point and
 * uvol_transcode_texture_segments_etc2_rgba take them as well - a plain ETC1 fit of each block's decoded texels (both flips, differential
 * or individual bases, best intensity table per half-block; not the basis transcoder's hint-driven path) and an EAC fit of their alpha,
 * gated by PSNR against the RGBA32 decode. */
int uvol_transcode_texture_segments_etc1(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments,
                                         uint8_t *const *blocks, size_t layer_cap, int outputs_on_device);

/* BC7 target (KTX2Loader's choice on desktop GPUs, reference src/lib/KTX2Loader.js:591-689): blocks[s * layers + l] receives
 * ceil(w/4) * ceil(h/4) 16-byte BC7 mode-6 blocks in raster order.  Lossy re-fit of the four ETC1S colours to 16 interpolation
 * weights (endpoint error <= 1, inner colours to the nearest weight); gated by PSNR against the RGBA32 decode, not bit parity. */
int uvol_transcode_texture_segments_bc7(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments,
                                        uint8_t *const *blocks, size_t layer_cap, int outputs_on_device);
/* UASTC sources (round 5): BC7 is what the stock loader asks a UASTC file for wherever ASTC is not supported - every desktop GPU
 * (reference src/lib/KTX2Loader.js:601-609, chosen at :665-676).  Single-plane blocks and solid blocks become mode 6 (each endpoint with
 * the p-bit that represents its four values best, every texel the nearest of the 16 interpolated colours), dual-plane blocks mode 5 with
 * the rotation that puts the second plane's channel into the scalar slot (UASTC's and BC7's 2-bit weights are equal).  A re-fit of the
 * block's own endpoints, gated by PSNR against the RGBA32 decode like the ETC1S case - the basis transcoder's tables are not restated. */
/* Files WITH alpha slices (images whose alpha was not 255; reference src/lib/KTX2Loader.js:493-497 reads them): the stock loader asks
 * such a file for the SECOND format of its table (:672-676) - BC7 with alpha or ETC2 RGBA.  uvol_transcode_texture_segments_bc7 then
 * writes mode-5 blocks whose alpha endpoints are the alpha block's lowest / highest level (exact) with 2-bit alpha indices;
 * uvol_transcode_texture_segments_etc2_rgba writes 16-byte ETC2_EAC RGBA8 blocks (EAC alpha block + the exact ETC1 colour re-pack;
 * opaque files get alpha 255).  Alpha is a re-fit (the basis transcoder's tables are not in the reference): gated by alpha PSNR against
 * the RGBA32 decode, like the BC7 colour.  The ETC1 target stays opaque-only and refuses a file with alpha slices. */
int uvol_transcode_texture_segments_etc2_rgba(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments,
                                              uint8_t *const *blocks, size_t layer_cap, int outputs_on_device);

/* ASTC 4x4 target for UASTC sources (KTX2Loader's first choice for them, reference src/lib/KTX2Loader.js:591-600, :648-689; BASELINE
 * configs[4] "KTX2 -> ASTC"): blocks[s * layers + l] receives ceil(w/4) * ceil(h/4) 16-byte ASTC blocks in raster order.  A direct
 * re-pack of the UASTC block (same endpoints and weights; the pair is swapped and the weights mirrored where ASTC would apply blue
 * contraction; solid blocks become void-extent blocks): an ASTC decoder reproduces the UASTC texels exactly.  ETC1S sources are
 * rejected with UVOL_E_UNSUPPORTED (the stock player never picks ASTC for them). */
int uvol_transcode_texture_segments_astc(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments,
                                         uint8_t *const *blocks, size_t layer_cap, int outputs_on_device);

/* ---- per-segment results for the batched texture calls (round 5; additive: uvol_abi_version stays 1) ----
 * The reference fails per `basisu` process (scripts/Encoder.py:293-298): one bad batch does not undo the others.  The batched texture
 * entry points above return ONE code for the call; these forms also fill status[s] per segment, like uvol_encode_mesh_batch does per
 * frame, and return UVOL_OK whenever the call itself ran (bad arguments, device errors and out-of-memory still fail the call).
 *
 * uvol_encode_texture_segments_st: as uvol_encode_texture_segments[_dev] (inputs_on_device selects which); a segment whose output buffer
 * is too small gets UVOL_E_NOSPACE (out_lens[s] = the size it needs), one the device could not encode UVOL_E_ENCODE; the others are written.
 *
 * uvol_transcode_texture_segments_st: every decode / transcode target through one entry point.  Files are judged one by one: an
 * unreadable container (UVOL_E_INVALID / UVOL_E_UNSUPPORTED from uvol_ktx2_info's rules), a shape other than the first readable file's
 * (UVOL_E_INVALID), a source kind the target does not take (UVOL_E_UNSUPPORTED: ASTC wants UASTC sources, ETC1 / ETC2 want ETC1S; RGBA32 and
 * BC7 take both), a payload that turns out corrupt on the device (UVOL_E_ENCODE) fail in their own slot; ETC1S and UASTC files may share a
 * batch.  out[s * layers + l] as in the entry point of the target; slots of failed segments are not read. */
enum { UVOL_TARGET_RGBA32 = 0, UVOL_TARGET_ETC1 = 1, UVOL_TARGET_BC7 = 2, UVOL_TARGET_ASTC = 3, UVOL_TARGET_ETC2_RGBA = 4,
       UVOL_TARGET_BC1 = 5, UVOL_TARGET_BC3 = 6 };
/* UVOL_TARGET_BC1 / UVOL_TARGET_BC3 (round 5, ETC1S sources, through uvol_transcode_texture_segments_st only): the stock loader's
 * `dxtSupported` row (reference src/lib/KTX2Loader.js:610-618: TranscoderFormat.BC1 for an opaque file, BC3 for one with alpha slices).
 * 8-byte BC1 blocks (four-colour mode; a file with alpha slices is UVOL_E_UNSUPPORTED: it is asked for BC3) / 16-byte BC3 blocks (BC4
 * alpha block from the alpha slice - 255 for an opaque file -, then the colour block), raster order.  Re-fits of the four ETC1S colours /
 * alpha levels, gated by PSNR against the RGBA32 decode like the BC7 target.  UASTC sources take both targets as well: the decoded texels of
 * a block are range-fitted (bounding-box corners along the sign of the R-G / B-G covariances, RGB565, nearest palette entry per texel), BC3
 * takes its BC4 alpha block from the texels' alpha, BC1 drops alpha. */
int uvol_encode_texture_segments_st(uvol_ctx *ctx, const uint8_t *const *rgba, int n_segments, int n_layers,
                                    uint32_t width, uint32_t height, int inputs_on_device,
                                    uint8_t *const *outs, const size_t *caps, size_t *out_lens, int *status);
int uvol_transcode_texture_segments_st(uvol_ctx *ctx, const uint8_t *const *ktx2, const size_t *lens, int n_segments,
                                       uint8_t *const *out, size_t layer_cap, int outputs_on_device, int target, int *status);

/* ---- decode path, geometry half ----
 * Replaces what the stock player obtains from the draco WASM decoder per frame (reference src/V2/player.ts:101, :313-336):
 * Draco 2.2 TRIANGULAR_MESH / valence-edgebreaker files with the attribute decoders of the fixtures (position, tex-coord,
 * normal, optional generic) -> de-quantised values per attribute in decoding order + one entry index per corner, i.e. the
 * inverse of uvol_mesh. */
typedef struct uvol_decoded_mesh {
  /* in: capacities of the caller's buffers */
  uint32_t cap_faces;              /* idx_* hold 3 * cap_faces entries */
  size_t cap_values;               /* pos / nrm hold 3 * cap_values floats, uv 2 * cap_values (3 * n_faces always suffices) */
  /* in: caller buffers (any may be NULL to skip that output) */
  float *pos, *uv, *nrm;
  uint32_t *idx_pos, *idx_uv, *idx_nrm;
  /* out */
  uint32_t n_faces, n_pos, n_uv, n_nrm;
} uvol_decoded_mesh;
/* host-only: face count of a .drc and the value count that always suffices (UVOL_E_INVALID for foreign data) */
int uvol_drc_info(const uint8_t *drc, size_t len, uint32_t *n_faces, uint32_t *max_values);
/* n frames, one kernel launch per stage; status[i] per frame (may be NULL).  The arrays of `out` are caller-owned host memory: pageable arrays
 * are filled through the library's pinned double buffers (host threads copy out of them); arrays that ALL lie in uvol_host_alloc memory are
 * written by the DMA engines where they are (round 6: no staging, no host copy of the 10.5 MB a 100 k-vertex frame decodes to). */
int uvol_decode_mesh_batch(uvol_ctx *ctx, const uint8_t *const *drc, const size_t *lens, int n, uvol_decoded_mesh *out, int *status);
/* Same with the buffers of `out` in HBM (device pointers): the decoded arrays stay on the device - what a GPU-resident consumer
 * (a renderer's vertex buffers, or uvol_encode_mesh_batch_dev[_out] re-encoding them) reads without a host round trip.  The .drc
 * files themselves are host memory; counts come back in `out`. */
int uvol_decode_mesh_batch_dev(uvol_ctx *ctx, const uint8_t *const *drc, const size_t *lens, int n, uvol_decoded_mesh *out, int *status);

/* ---- measurement hooks (bench.py / rocprof cross-check) ---- */
/* When enabled, every kernel group is bracketed by hipEvents on the ctx stream. */
int uvol_profile_enable(uvol_ctx *ctx, int on);
int uvol_profile_reset(uvol_ctx *ctx);
/* Number of distinct kernel groups recorded so far. */
int uvol_profile_count(uvol_ctx *ctx);
/* name/launch-count/total-ms/algorithmic-bytes of group i (for the matrix-core sub-group tex.k10_sel_assign the last field
 * counts integer multiply/add operations instead of bytes). */
int uvol_profile_get(uvol_ctx *ctx, int i, char *name, size_t name_cap,
                     uint64_t *launches, double *total_ms, uint64_t *algo_bytes);

#ifdef __cplusplus
}
#endif
#endif /* UVOL_CODEC_H */
