/* crt_scene_io.h — C API of the harness's scene-file readers (libcrt_scene_io.so; plain C++, no HIP).
 *
 * the step before set_scene. The reference reads OBJ through tinyobjloader inside
 * Scene::load_obj (util/scene.cpp:94-228); this is the text half of that -- groups, tinyobjloader's ear-clipping triangulation, re-indexing on
 * unique (position, normal, uv) index triples in order of first use, the material name in force at a group's first
 * face -- as a streaming reader that takes a 10 M-triangle file in seconds. Materials, textures, the generated light
 * and the Scene assembly stay in chameleonrt_amd/obj_io.py (scene.cpp:191-227), which binds these entry points with
 * ctypes; a ChameleonRT build would keep its own importer and hand the Scene to crt_hip_set_scene (include/crt_hip.h).
 */
#ifndef CRT_SCENE_IO_H
#define CRT_SCENE_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct crt_obj_file crt_obj_file;

/* Never NULL; crt_obj_error() tells whether the file could be read and parsed (NULL = fine). */
crt_obj_file *crt_obj_parse(const char *path);
const char *crt_obj_error(const crt_obj_file *f);
void crt_obj_free(crt_obj_file *f);

int crt_obj_num_mtllibs(const crt_obj_file *f);
const char *crt_obj_mtllib(const crt_obj_file *f, int i);

/* One shape per `o` / `g` group that has faces (tinyobj emits no empty shapes), in file order. has_uv: 1 / 0, or -1 if
 * the group mixes vertices with and without texture coordinates (an error for the importer). has_material: a usemtl
 * was in force at the group's first face; crt_obj_shape_material then names it. */
int crt_obj_num_shapes(const crt_obj_file *f);
int crt_obj_shape_info(const crt_obj_file *f, int shape, uint64_t *n_vertices, uint64_t *n_triangles, int *has_uv,
                       int *has_material);
const char *crt_obj_shape_material(const crt_obj_file *f, int shape);
/* how many mtllib lines had been read when that usemtl was met: the name resolves against the materials of those only
 * (tinyobj looks a material up when it parses the usemtl line) */
int crt_obj_shape_material_libs(const crt_obj_file *f, int shape);
/* vertices: n_vertices * 3 floats, indices: n_triangles * 3, uvs: n_vertices * 2 floats (ignored unless has_uv == 1) */
int crt_obj_shape_copy(const crt_obj_file *f, int shape, float *vertices, uint32_t *indices, float *uvs);

/* ---- texture files ------------------------------------------------------------------------------------------------
 * JPEG decoding with the arithmetic of the reference's decoder (its vendored stb_image, util/material.cpp:5-17 ->
 * util/stb_image.h: inverse DCT, chroma up-sampling and YCbCr -> RGB are implementation choices on which libjpeg-based
 * decoders differ from it by up to 2/255). bytes: the file's contents. On success (0) *rgba points to width * height * 4
 * bytes, rows top to bottom, alpha 255 -- what stbi_load(..., 4) returns BEFORE the reference's vertical flip -- to be
 * released with crt_image_free. Baseline and progressive, 8-bit, 1 / 3 / 4 components, restart intervals. */
int crt_image_decode_jpeg(const uint8_t *bytes, uint64_t n_bytes, int32_t *width, int32_t *height, uint8_t **rgba);
void crt_image_free(uint8_t *rgba);
const char *crt_image_error(void);

#ifdef __cplusplus
}
#endif
#endif /* CRT_SCENE_IO_H */
