/* rt_mi355x.h — C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * Drop-in boundary for the render path of nabilbrice/raytracer: the Rust crate
 * keeps Config / SetupCamera / Camera / Shape / Material and its PPM writer and
 * calls these entry points in place of raytracer::accel_render_into_file
 * (src/lib.rs:159-184) and raytracer::render_into_file (src/lib.rs:137-157).
 * Plain C: POD structs, pointers and sizes only.  Every function that can fail
 * returns an int status (RT_OK = 0) and never unwinds across the boundary; the
 * reference signals the same conditions with panics.  rt_last_error() gives the
 * message of the calling thread's last failure.
 *
 * A scene handle is not thread-safe: one call at a time per handle.  Renders of one handle also run one at a
 * time on the device, whatever streams they are given (they share the handle's work counter and partial sums:
 * each render waits for the handle's previous one).  Every entry point leaves the calling thread's current HIP
 * device as it found it.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = 1, /* null pointer, zero resolution, empty sample range, bad index ... */
    RT_ERR_UNSUPPORTED = 2,      /* shape / material outside src/geometry.rs:9-15, src/materials.rs:15-35 */
    RT_ERR_DEVICE = 3,           /* no gfx950 device, HIP call failed */
    RT_ERR_IO = 4,               /* file could not be opened / written / parsed */
    RT_ERR_LIMIT = 5             /* size beyond what an entry point supports */
};

enum rt_shape_kind { RT_SHAPE_SPHERE = 0, RT_SHAPE_DISC = 1 }; /* src/geometry.rs:9-15 (BoundVolume is serde-skipped) */
enum rt_material_kind {                                        /* src/materials.rs:15-35 */
    RT_MAT_DIFFUSE = 0,
    RT_MAT_METAL = 1,
    RT_MAT_DIELECTRIC = 2,
    RT_MAT_TEXTUREMAP = 3,
    RT_MAT_EMITTER = 4
};

/* src/camera.rs:4-16, field for field. */
typedef struct rt_camera {
    double lookat[3];
    double lookfrom[3];
    double lookup[3];
    double focal_distance;
    double inv_focal_length;
    double aperture;
    double horiz_arm[3];
    double vert_arm[3];
    uint32_t horiz_res;
    uint32_t vert_res;
    double aspect_ratio; /* computed and never used, as in src/camera.rs:33 */
} rt_camera;

/* src/lib.rs:25-29 Hittable {shape, material}, flattened.  The primitive ID of a
 * hittable is its index in the array handed to rt_scene_create (the order of
 * scene.push in src/scenegen.rs:23-61, or of "hittables" in scene.json). */
typedef struct rt_hittable {
    int32_t shape;           /* rt_shape_kind */
    int32_t material;        /* rt_material_kind */
    double centre[3];        /* src/geometry.rs:38, :44 */
    double normal[3];        /* disc only; used as given (serde bypasses Disc::new, src/geometry.rs:42-47) */
    double radius;           /* src/geometry.rs:39, :46 */
    double albedo[3];        /* Diffuse, Metal, Emitter */
    double fuzz;             /* Metal */
    double refractive_index; /* Dielectric */
    double orient_up[3];     /* TextureMap */
    double orient_around[3]; /* TextureMap */
    int32_t texture;         /* TextureMap: index into the texture table, else -1 */
    int32_t reserved;
} rt_hittable;

/* Decoded image of a TextureMap ("map" in src/materials.rs:26-31): 8-bit RGB,
 * row-major, row 0 = top, no padding. */
typedef struct rt_texture {
    const uint8_t* rgb8;
    uint32_t width;
    uint32_t height;
} rt_texture;

typedef struct rt_scene rt_scene; /* opaque: flattened BVH + SoA primitives resident in HBM */

typedef struct rt_scene_info {
    uint64_t n_hittables, n_spheres, n_discs;
    uint64_t n_nodes;      /* inner nodes of the flattened two-child BVH (leaves live in their parent) */
    uint32_t bvh_depth;    /* inner-node levels */
    uint32_t stack_depth;  /* traversal stack entries per lane compiled for this scene */
    uint64_t device_bytes; /* HBM held by the scene */
    double build_seconds;  /* host BVH build + flatten + upload */
    int32_t device;
    int32_t lds_resident;  /* 1 when nodes and spheres are staged in LDS by the render kernel */
    uint32_t n_wide;       /* spheres kept out of the BVH because their box covers most of the others (tested once
                              per segment, like the discs); n_nodes counts the tree over the remaining spheres */
    uint32_t n_shells;     /* of the n_wide spheres, those that form a group of nested shells (concentric, radii well
                              apart): the render kernels test only the three of them a segment can meet first; 0 = no
                              such group */
} rt_scene_info;

/* Sample range and launch options of one render call. */
typedef struct rt_render_opts {
    uint32_t sample_begin; /* first sample index of every pixel traced by this call */
    uint32_t sample_end;   /* one past the last; sample_end - sample_begin paths per pixel */
    uint32_t depth;        /* scatter_depth of src/lib.rs:88 (the reference passes 10, src/lib.rs:172) */
    uint32_t flags;        /* RT_RENDER_* */
    uint64_t seed;         /* Philox4x32-10 key; counter = (pixel, sample, bounce, attempt) */
    uint32_t samples_per_item; /* a pixel's samples are traced in chunks of this many by whichever lane is free; a chunk
                                  is summed in sample order, the chunks are added in chunk order.  0 = chosen from
                                  the frame size (rt_render_chunk reports it) */
    uint32_t reserved;
} rt_render_opts;
enum {
    RT_RENDER_ACCUMULATE = 1,          /* add to the buffer instead of overwriting it */
    RT_RENDER_KERNEL_LOCKSTEP = 2,     /* measurement aid: the plain one-segment-per-turn kernel */
    RT_RENDER_KERNEL_NO_LDS_SCENE = 4, /* measurement aid: keep the BVH in HBM even when it would fit in LDS */
    RT_RENDER_STATS = 8,               /* measurement aid: instrumented kernel, see rt_debug_render_stats */
    RT_RENDER_TIMING = 16,             /* measurement aid: production kernel with a cycle stamp around each phase */
    RT_RENDER_DEFER_LEAF = 32,         /* measurement aid, with RT_RENDER_STATS: a lane puts its first leaf off and descends on */
    RT_RENDER_KERNEL_WAVEFRONT = 64    /* the render as a pipeline of kernels (generate, then trace + shade per bounce) whose rays
                                          meet in queues in device memory, compacted by the material they hit; same image bit
                                          for bit, several times slower than the default persistent kernel; allocates a path pool
                                          (104 bytes per path slot, at most 2^25 slots) on first use */
};

const char* rt_last_error(void);

/* Version of this ABI (major << 16 | minor). */
uint32_t rt_abi_version(void);

/* Number of usable gfx950 devices; 0 when there is none. */
int rt_device_count(void);

/* src/camera.rs:19-47 Camera::build. */
int rt_camera_build(const double lookat[3], const double lookfrom[3], double inv_focal_length, double aperture,
                    uint32_t horiz_res, uint32_t vert_res, rt_camera* out);

/* src/scenegen.rs:70-79 default_camera with the resolution as arguments (the reference fixes 512 x 512). */
int rt_default_camera(uint32_t horiz_res, uint32_t vert_res, rt_camera* out);

/* src/scenegen.rs:12-109 gen_scene: 488 hittables in push order.  The reference
 * draws from an unseeded ThreadRng; here a SplitMix64 stream seeded by `seed`
 * is used in the same draw order.  Writes at most cap entries, *n = 488. */
int rt_gen_scene(uint64_t seed, rt_hittable* out, size_t cap, size_t* n);

/* Build-defined stress scenes made only of the reference's shapes and materials.
 * grid: ground sphere + side*side small spheres (r = 0.2, pitch 0.5, side <= 2800) resting on it, the
 * material mix of gen_scene.  shells: the scene.json earth sphere (texture 0) +
 * `shells` concentric dielectric spheres with alternating index 1.5 / (1/1.5). */
int rt_gen_scene_grid(uint64_t seed, uint32_t side, rt_hittable* out, size_t cap, size_t* n);
int rt_gen_scene_shells(uint32_t shells, rt_hittable* out, size_t cap, size_t* n);

/* Box<[Hittable]> / Box<CoveringTree> of src/main.rs:10-13: copies the inputs,
 * builds and flattens the BVH (replaces make_coveringtree,
 * src/boundingvolume.rs:192-208) and uploads everything to `device`.
 * Discs are kept outside the BVH and tested one by one, as src/lib.rs:46-49 does. */
int rt_scene_create(const rt_hittable* hittables, size_t n_hittables, const rt_texture* textures, size_t n_textures,
                    int device, rt_scene** out);

/* The same with a choice of tree builder.  Both replace make_coveringtree (src/boundingvolume.rs:192-208) and
 * give the same nearest hits; they differ in build time and in how many boxes a ray visits.
 *   RT_BUILDER_SAH_HOST     binned surface-area heuristic on the host, large subtrees on separate threads (default; 1 M spheres: ~0.4 s)
 *   RT_BUILDER_LBVH_DEVICE  Morton-order linear BVH built on the GPU (1 M spheres: ~0.25 s with the host-side tables, slower to trace) */
enum rt_builder { RT_BUILDER_DEFAULT = 0, RT_BUILDER_SAH_HOST = 1, RT_BUILDER_LBVH_DEVICE = 2 };
typedef struct rt_scene_opts {
    uint32_t builder; /* rt_builder */
    uint32_t reserved[7];
} rt_scene_opts;
int rt_scene_create_ex(const rt_hittable* hittables, size_t n_hittables, const rt_texture* textures,
                       size_t n_textures, int device, const rt_scene_opts* opts, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out);

/* accel_render_into_file / render_into_file (src/lib.rs:137-184) without the
 * file: traces samples [sample_begin, sample_end) of every pixel and leaves the
 * per-pixel SUM of the path colours, 3 floats per pixel, row-major with row 0 at
 * the top (src/lib.rs:166-167), in d_sum_rgb — DEVICE memory of the scene's
 * device.  `stream` is a hipStream_t (NULL = default stream); the call is
 * asynchronous with respect to the host: it enqueues and returns.  The one
 * exception: a render in chunks (rt_render_chunk) whose partial sums do not fit
 * the scene's buffer waits for the handle's previous render and grows the buffer
 * first; rt_scene_reserve does that ahead of time. */
int rt_render_device(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, float* d_sum_rgb,
                     void* stream);

/* Sizes the scene's partial-sum buffer for renders of this frame and sample range (and anything smaller), so that
 * rt_render_device neither allocates nor waits.  No counterpart in the reference (it has no device). */
int rt_scene_reserve(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts);
/* Upper bound of that buffer: when the library chooses the chunking (samples_per_item = 0) a render is cut into at
 * most RT_PARTIALS_BUDGET_BYTES / (horiz_res * vert_res * 12) chunks (and never more than 64), so the partial sums of
 * any frame stay within this budget (a 3840x2160 frame: 10 chunks; chunks grow longer instead).  A chunking given
 * by the caller is taken as it is. */
#define RT_PARTIALS_BUDGET_BYTES (1024u * 1024u * 1024u)

/* rt_render_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same result; the
 * work counter and the partial sums of a chunked render live in a fixed array of RT_NOALLOC_SCRATCH_BYTES of
 * device memory that is part of the library's code object (one copy per GPU, held until the process ends), so the
 * call makes no hipMalloc / hipFree.  Limits: with samples_per_item given, the partial sums
 * (chunks * horiz_res * vert_res * 12 bytes, chunks > 1) must fit RT_NOALLOC_SCRATCH_BYTES - 2048, else RT_ERR_LIMIT;
 * with samples_per_item = 0 the library uses min(the chunk count of rt_render_chunk, what fits, at least 1), i.e.
 * samples per item = ceil(spp / that count), so large frames are summed in fewer, longer chunks than
 * rt_render_device would use (same image up to f32 summation order).  Every call on a GPU uses that GPU's one array:
 * calls that overlap on one GPU — different scene handles on different streams — race; calls through one handle
 * are ordered like all renders of a handle.  RT_RENDER_STATS / RT_RENDER_TIMING are refused here. */
#define RT_NOALLOC_SCRATCH_BYTES (64u * 1024u * 1024u)
int rt_render_device_noalloc(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, float* d_sum_rgb,
                             void* stream);

/* Measurement aid: phase counters of the last RT_RENDER_STATS render on this scene:
 * {inner-node turns*64, lanes in them, leaf turns*64, lanes, shade turns*64, lanes shading or
 * regenerating, lanes regenerating, descent-loop turns*64}; after an RT_RENDER_TIMING render the same
 * words hold {shade, descent-loop header, inner loop, leaf turn, whole kernel} wave cycles summed over
 * waves, the number of waves, 0, 0.  Synchronises the device. */
int rt_debug_render_stats(rt_scene* scene, uint64_t out[8]);

/* Parity aid: the render kernel itself (the production code, assembly loops included, built with one extra
 * store per path) renders samples [sample_begin, sample_end) of every pixel and returns the colour of each
 * path, rgb[(pixel * spp + sample - sample_begin) * 3 ..] — what accel_raytrace (src/lib.rs:88-135) returns
 * for that camera ray.  Host buffer of horiz_res * vert_res * spp * 3 floats; synchronous. */
int rt_debug_render_paths(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, float* rgb);

/* Measurement aid: how rt_render_device launches its kernel for this scene on this device: threads per workgroup,
 * workgroups the runtime can keep resident per CU (registers and LDS), and the waves per SIMD that follow. */
int rt_debug_render_launch(rt_scene* scene, uint32_t* block_threads, uint32_t* workgroups_per_cu, uint32_t* waves_per_simd);

/* Chunk size rt_render_device uses for this scene, frame and sample range (the summation order of the image): a
 * scene whose tree is read from HBM is rendered in chunks of two samples, one that sits in LDS in longer ones. */
int rt_render_chunk(const rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, uint32_t* samples_per_item);

/* (1/spp) * sum, then color_to_ppm (src/lib.rs:148-149, :186-192): device
 * buffers, n_pixels*3 channels. */
int rt_finalize_device(const rt_scene* scene, const float* d_sum_rgb, size_t n_pixels, uint32_t spp_total,
                       uint8_t* d_rgb8, void* stream);

/* The same on host arrays, without a device (used when a finished checkpoint only has to become an image). */
int rt_finalize(const float* sum_rgb, size_t n_pixels, uint32_t spp_total, uint8_t* rgb8);

/* Host-buffer convenience over the two calls above: sum_rgb (optional) and rgb8
 * (optional) are HOST arrays of horiz_res*vert_res*3; synchronous.  With
 * RT_RENDER_ACCUMULATE sum_rgb is read first and taken to hold samples
 * [0, sample_begin) of every pixel; rgb8 is then the image of all sample_end samples. */
int rt_render(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, float* sum_rgb, uint8_t* rgb8);

/* One render over several GPUs of the node from ONE process — what a host that cannot start a process per GPU
 * (the Rust crate calling through FFI) uses in place of the per-rank path.  `scenes` are handles of the same
 * hittables created on different devices (rt_scene_create(..., device = g, ...)); handle g traces the sample
 * indices [begin + g*spp/G, begin + (g+1)*spp/G) of every pixel on its device, all devices at the same time; the
 * partial sums then meet on the first handle's device, which applies mean / gamma / quantise: by one RCCL
 * ncclReduce(sum) over xGMI when every handle sits on a different device (librccl.so is bound at the first such
 * call), else — or with RT_MULTI_REDUCE=peer in the environment — by peer-to-peer copies, all in flight at once,
 * into landing slots on the first device that one kernel adds in handle order.  The image is that of one device
 * tracing all samples up to f32 summation order.  Host buffers as in rt_render; synchronous. */
int rt_render_multi(rt_scene* const* scenes, size_t n_scenes, const rt_camera* cam, const rt_render_opts* opts,
                    float* sum_rgb, uint8_t* rgb8);

/* Nearest hit of n rays: what tree_filter + min_by produce for one segment
 * (src/boundingvolume.rs:214-231, src/lib.rs:94-98; the linear scan of
 * src/lib.rs:46-49 when the scene holds discs).  orig/dir: n*3 floats, dir is
 * used as given.  id[k] = primitive ID or -1; t[k] and normal[3k..] are written
 * for hits only.  Equal parameters resolve to the lowest ID.  Host buffers. */
int rt_trace_primary(rt_scene* scene, const float* orig, const float* dir, size_t n, int32_t* id, double* t,
                     double* normal);

/* Camera rays the render kernel starts from (src/camera.rs:49-78 + src/ray.rs:9-11)
 * for pixels [pixel_begin, pixel_end) (index j*horiz_res+i) and samples
 * [sample_begin, sample_end); ray r = (pixel-pixel_begin)*ns + (sample-sample_begin).
 * Host buffers of n*3 floats. */
int rt_generate_primary(rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t pixel_begin,
                        uint32_t pixel_end, uint32_t sample_begin, uint32_t sample_end, float* orig, float* dir);

/* Colour of single paths: rgb[3k..] = accel_raytrace of the camera ray of
 * (pixel[k], sample[k]) (src/lib.rs:88-135).  Host buffers. */
int rt_trace_paths(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, const uint32_t* pixel,
                   const uint32_t* sample, size_t n, float* rgb);

/* The same paths, segment by segment: what the loop of src/lib.rs:93-131 sees on each turn.
 * seg_id[k*max_segments + i] / seg_t[...] = primitive ID (-1 = the miss that ends the path) and
 * parameter of segment i of path k, for i < min(n_seg[k], max_segments); n_seg[k] = segments traced.
 * rgb is optional here.  Host buffers. */
int rt_trace_path_segments(rt_scene* scene, const rt_camera* cam, const rt_render_opts* opts, const uint32_t* pixel,
                           const uint32_t* sample, size_t n, float* rgb, uint32_t max_segments, int32_t* seg_id,
                           double* seg_t, uint32_t* n_seg);

/* src/main.rs:35-42 + src/lib.rs:151: "P3\nW H\n255\n" then one "r g b\n" line
 * per pixel.  Unlike the reference the file is truncated first. */
int rt_write_ppm(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);
/* The same image as a binary PPM: "P6\nW H\n255\n" (the header of src/main.rs:41-42 with the binary magic number)
 * followed by W*H*3 raw bytes, rows top to bottom — a third of the size of the text form of src/lib.rs:151 and what
 * image tools read fastest.  The reference writes P3 only. */
int rt_write_ppm_p6(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

/* ---- the host side of src/main.rs: scene file, texture images, file output ------------------- */

/* Config of src/config.rs:7-11 read from a scene file (src/main.rs:26-32): "camera" is a SetupCamera
 * (src/config.rs:13-21) turned into a Camera by SetupCamera::setup (src/config.rs:24-27), "hittables" a
 * list of {"shape": {"Sphere"|"Disc": {...}}, "material": {"Diffuse"|"Metal"|"Dielectric"|"TextureMap"|
 * "Emitter": {...}}} in serde's externally tagged form; Vec3 is a 3-array, Color an object {r, g, b}.
 * "map" of a TextureMap is a PATH whose image is decoded while the scene is read (src/materials.rs:37-46);
 * a relative path is taken relative to texture_dir (rt_config_load: the directory of the scene file).
 * Unknown keys are ignored, missing keys and wrong types are errors, as with serde's derive.  Where the
 * reference panics ("unable to read scene file", "unable to deserialize scene information", "cannot open
 * file") these return RT_ERR_IO with that text in rt_last_error(). */
typedef struct rt_config rt_config;
int rt_config_load(const char* path, rt_config** out);
int rt_config_parse(const char* json, size_t len, const char* texture_dir, rt_config** out);
void rt_config_destroy(rt_config* cfg);
/* Borrowed views into the config (valid until rt_config_destroy); any output pointer may be NULL. */
int rt_config_view(const rt_config* cfg, rt_camera* camera, const rt_hittable** hittables, size_t* n_hittables,
                   const rt_texture** textures, size_t* n_textures);

/* image::open of src/materials.rs:37-39 for the one format the reference's scenes use: decodes a baseline
 * (or extended-sequential Huffman) 8-bit JPEG with 1 or 3 components into RGB8, row 0 = top.  *rgb8 is
 * allocated by the library; release it with rt_free. */
int rt_decode_jpeg(const uint8_t* bytes, size_t len, uint8_t** rgb8, uint32_t* width, uint32_t* height);
int rt_load_jpeg(const char* path, uint8_t** rgb8, uint32_t* width, uint32_t* height);
void rt_free(void* p);

/* render_into_file (src/lib.rs:137-157) and accel_render_into_file (src/lib.rs:159-184): render `spp`
 * samples per pixel and append one "r g b\n" line per pixel, rows top to bottom, to the open file
 * descriptor `fd`, which the caller has positioned after the PPM header (src/main.rs:41-42) — the
 * reference takes the open File the same way.  `depth` is the literal 10 of src/lib.rs:145 / :172 made
 * an argument; `seed` keys the sampler.  The scene is built, used and dropped inside the call, as the
 * reference drops its tree at return (src/lib.rs:162).  The accel variant accepts spheres only: the
 * reference's cover construction (src/boundingvolume.rs:242-259) has no arm for a Disc. */
int rt_render_into_file(int fd, const rt_camera* cam, const rt_hittable* hittables, size_t n_hittables,
                        const rt_texture* textures, size_t n_textures, uint32_t spp, uint32_t depth, uint64_t seed,
                        int device);
int rt_accel_render_into_file(int fd, const rt_camera* cam, const rt_hittable* hittables, size_t n_hittables,
                              const rt_texture* textures, size_t n_textures, uint32_t spp, uint32_t depth, uint64_t seed,
                              int device);
/* The two calls above for a binary file: the caller writes "P6\nW H\n255\n" where src/main.rs:41-42 writes the P3
 * header, and the pixels are appended as three raw bytes each in place of the "r g b\n" lines of src/lib.rs:151. */
int rt_render_into_file_p6(int fd, const rt_camera* cam, const rt_hittable* hittables, size_t n_hittables,
                           const rt_texture* textures, size_t n_textures, uint32_t spp, uint32_t depth, uint64_t seed,
                           int device);
int rt_accel_render_into_file_p6(int fd, const rt_camera* cam, const rt_hittable* hittables, size_t n_hittables,
                                 const rt_texture* textures, size_t n_textures, uint32_t spp, uint32_t depth, uint64_t seed,
                                 int device);

/* Progressive renders.  The reference renders all samples in one go and keeps nothing (src/lib.rs:137-157); with
 * the counter-based sampler a render can stop after any sample index and continue later: a checkpoint file holds
 * the running per-pixel sums (what rt_render leaves in sum_rgb) and what produced them.  Continuing with
 * RT_RENDER_ACCUMULATE from samples_done gives the image of an uninterrupted render (bit for bit when both use
 * the same rt_render_opts.samples_per_item and samples_done is a multiple of it, else to f32 rounding).
 * `tag` is the caller's to choose (the CLI stores a hash of camera and hittables) and is returned by the load.
 * rt_checkpoint_load with sum_rgb == NULL reads the header only; capacity counts floats. */
typedef struct rt_checkpoint_info {
    uint32_t width, height;
    uint32_t samples_done; /* the sums hold samples [0, samples_done) */
    uint32_t depth;
    uint64_t seed;
    uint64_t tag;
} rt_checkpoint_info;
int rt_checkpoint_save(const char* path, const rt_checkpoint_info* info, const float* sum_rgb);
int rt_checkpoint_load(const char* path, rt_checkpoint_info* info, float* sum_rgb, size_t capacity);
/* FNV-1a over a byte range, chained through `basis` (0 = start): a convenient `tag`. */
uint64_t rt_hash_bytes(const void* bytes, size_t len, uint64_t basis);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
