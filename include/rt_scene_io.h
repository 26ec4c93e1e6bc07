/*
 * rt_scene_io.h — byte-for-byte file form of an rt_scene (header-only, C and C++).
 *
 * Layout: magic "RTSCENE2", eight int64 counts, rt_settings, then the arrays in
 * the order volumes, meshes, tri_pos, tri_nrm, materials, lights, cameras,
 * animations, keyframes.
 * Native endianness and struct layout (x86-64 little endian on every box this
 * runs on).  rt_scene_read allocates with malloc; release with rt_scene_free.
 */
#ifndef RT_SCENE_IO_H
#define RT_SCENE_IO_H

#include "rt_scene.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline int rt_scene_write(const rt_scene* scene, const char* path)
{
    FILE* f = fopen(path, "wb");
    if (!f)
    {
        return -1;
    }
    const char magic[8] = {'R', 'T', 'S', 'C', 'E', 'N', 'E', '2'};
    int64_t counts[8] = {scene->volume_count,   scene->mesh_count,  scene->triangle_count,  scene->material_count,
                         scene->light_count,    scene->camera_count, scene->animation_count, scene->keyframe_count};
    int ok = 1;
    ok &= fwrite(magic, 1, 8, f) == 8;
    ok &= fwrite(counts, sizeof(int64_t), 8, f) == 8;
    ok &= fwrite(&scene->settings, sizeof(rt_settings), 1, f) == 1;
#define RT_IO_PUT(ptr, type, n) \
    if ((n) > 0) ok &= fwrite((ptr), sizeof(type), (size_t)(n), f) == (size_t)(n)
    RT_IO_PUT(scene->volumes, rt_volume, scene->volume_count);
    RT_IO_PUT(scene->meshes, rt_mesh, scene->mesh_count);
    RT_IO_PUT(scene->tri_pos, double, 9 * scene->triangle_count);
    RT_IO_PUT(scene->tri_nrm, double, 9 * scene->triangle_count);
    RT_IO_PUT(scene->materials, rt_material, scene->material_count);
    RT_IO_PUT(scene->lights, rt_light, scene->light_count);
    RT_IO_PUT(scene->cameras, rt_camera, scene->camera_count);
    RT_IO_PUT(scene->animations, rt_animation, scene->animation_count);
    RT_IO_PUT(scene->keyframes, rt_keyframe, scene->keyframe_count);
#undef RT_IO_PUT
    fclose(f);
    return ok ? 0 : -2;
}

static inline void rt_scene_free(rt_scene* scene)
{
    if (!scene)
    {
        return;
    }
    free(scene->volumes);
    free(scene->meshes);
    free(scene->tri_pos);
    free(scene->tri_nrm);
    free(scene->materials);
    free(scene->lights);
    free(scene->cameras);
    free(scene->animations);
    free(scene->keyframes);
    memset(scene, 0, sizeof(*scene));
}

static inline int rt_scene_read(rt_scene* scene, const char* path)
{
    memset(scene, 0, sizeof(*scene));
    FILE* f = fopen(path, "rb");
    if (!f)
    {
        return -1;
    }
    char magic[8];
    int64_t counts[8];
    int ok = 1;
    ok &= fread(magic, 1, 8, f) == 8;
    ok &= ok && memcmp(magic, "RTSCENE2", 8) == 0;
    ok &= ok && fread(counts, sizeof(int64_t), 8, f) == 8;
    ok &= ok && fread(&scene->settings, sizeof(rt_settings), 1, f) == 1;
    if (!ok)
    {
        fclose(f);
        return -2;
    }
    for (int i = 0; i < 8; ++i)
    {
        if (counts[i] < 0 || counts[i] > ((int64_t)1 << 32))
        {
            fclose(f);
            return -2;
        }
    }
    scene->volume_count = counts[0];
    scene->mesh_count = counts[1];
    scene->triangle_count = counts[2];
    scene->material_count = counts[3];
    scene->light_count = counts[4];
    scene->camera_count = counts[5];
    scene->animation_count = counts[6];
    scene->keyframe_count = counts[7];
#define RT_IO_GET(field, type, n)                                              \
    if ((n) > 0)                                                               \
    {                                                                          \
        scene->field = (type*)malloc(sizeof(type) * (size_t)(n));              \
        ok &= scene->field != NULL &&                                          \
              fread(scene->field, sizeof(type), (size_t)(n), f) == (size_t)(n); \
    }
    RT_IO_GET(volumes, rt_volume, scene->volume_count);
    RT_IO_GET(meshes, rt_mesh, scene->mesh_count);
    RT_IO_GET(tri_pos, double, 9 * scene->triangle_count);
    RT_IO_GET(tri_nrm, double, 9 * scene->triangle_count);
    RT_IO_GET(materials, rt_material, scene->material_count);
    RT_IO_GET(lights, rt_light, scene->light_count);
    RT_IO_GET(cameras, rt_camera, scene->camera_count);
    RT_IO_GET(animations, rt_animation, scene->animation_count);
    RT_IO_GET(keyframes, rt_keyframe, scene->keyframe_count);
#undef RT_IO_GET
    fclose(f);
    if (!ok)
    {
        rt_scene_free(scene);
        return -2;
    }
    return 0;
}

#endif /* RT_SCENE_IO_H */
