#include "FrameSetup.h"

#include "Math.h"

#include <algorithm>
#include <cmath>

FrameDerived deriveFrame(const rt_scene& scene)
{
    const rt_settings& s = scene.settings;
    FrameDerived d;
    d.cellSize = s.hash_grid_cell_size;
    if (scene.camera_count > 0 && s.image_height > 0 && scene.volume_count > 0)
    {
        const rt_camera& cam = scene.cameras[0];
        Vector centroid{0.0, 0.0, 0.0};
        for (int64_t i = 0; i < scene.volume_count; ++i)
        {
            const double* p = scene.volumes[i].position;
            centroid = centroid + Vector{p[0], p[1], p[2]};
        }
        centroid = centroid / static_cast<double>(scene.volume_count);
        const Vector eye{cam.position[0], cam.position[1], cam.position[2]};
        const double depth = (centroid - eye).magnitude();
        const double pixelHalfAngle = 0.5 * Utility::radians(cam.vertical_fov_degrees) /
                                      static_cast<double>(cam.height > 0 ? cam.height : s.image_height);
        const double representativeRadius = depth * std::tan(pixelHalfAngle);
        if (representativeRadius > 0.0)
        {
            d.cellSize = representativeRadius;
            d.sceneDepthFootprint = representativeRadius;
        }
    }
    d.minSplatRadius = (d.sceneDepthFootprint > 0.0 && s.splat_min_radius_scale > 0.0)
                           ? s.splat_min_radius_scale * d.sceneDepthFootprint
                           : 0.0;
    d.cellSize *= std::max(0.05, s.density_cell_scale);
    const double footprint = d.sceneDepthFootprint > 0.0 ? d.sceneDepthFootprint : d.cellSize;
    d.keepRadius = std::max(footprint, s.probe_keep_radius_scale * footprint);
    return d;
}
