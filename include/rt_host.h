/* rt_host.h — C view of the host-side helpers (pure C++17, no HIP) that sit
 * either side of the GPU hot path: the scene script + OFF loader + flattener
 * (reference source/Main.cpp:26-208, source/Mesh.h:45-90), the background /
 * PPM writer (source/Image.cpp:12-43) and the photon kd-tree ORDER builder
 * (source/kdtree.h:60-69).  Exposed so that Python tests / bench.py build
 * exactly the scenes the C++ application builds.  Library: librt_host.so. */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_host_scene rt_host_scene;

/* kind: cubes | lowres | hires | stress | file:<name.off>  (ScenePresets.h) */
int rt_host_scene_build(const char* kind, const char* mesh_dir, uint32_t width, uint32_t height,
                        rt_host_scene** out);
const rt_scene_desc* rt_host_scene_desc(const rt_host_scene* s);
void rt_host_scene_free(rt_host_scene* s);
const char* rt_host_last_error(void);

/* Image::fillBackground / Image::savePPM on a raw [h][w][3] float buffer. */
void rt_host_fill_background(float* rgb, uint32_t width, uint32_t height);
int rt_host_save_ppm(const char* path, const float* rgb, uint32_t width, uint32_t height);

/* kdtree::make_tree: permute n photons (pos[n][3], dir[n][3], weight[n]) in
 * place into the median-implicit order the k-NN kernel walks. */
int rt_host_kd_order(float* pos3, float* dir3, float* weight, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
