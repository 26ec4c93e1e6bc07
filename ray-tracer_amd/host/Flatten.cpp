#include "Flatten.h"

#include <cstring>
#include <map>

namespace
{

void put3(double* dst, const Vector& v)
{
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

void put4(double* dst, const Quaternion& q)
{
    dst[0] = q.x;
    dst[1] = q.y;
    dst[2] = q.z;
    dst[3] = q.w;
}

void putName(char* dst, const std::string& name)
{
    std::memset(dst, 0, 64);
    std::strncpy(dst, name.c_str(), 63);
}

}  // namespace

void FlatScene::bind()
{
    scene.volume_count = static_cast<int64_t>(volumes.size());
    scene.mesh_count = static_cast<int64_t>(meshes.size());
    scene.triangle_count = static_cast<int64_t>(triPos.size() / 9);
    scene.material_count = static_cast<int64_t>(materials.size());
    scene.light_count = static_cast<int64_t>(lights.size());
    scene.camera_count = static_cast<int64_t>(cameras.size());
    scene.volumes = volumes.data();
    scene.meshes = meshes.data();
    scene.tri_pos = triPos.data();
    scene.tri_nrm = triNrm.data();
    scene.materials = materials.data();
    scene.lights = lights.data();
    scene.cameras = cameras.data();
    scene.animation_count = static_cast<int64_t>(animations.size());
    scene.keyframe_count = static_cast<int64_t>(keyframes.size());
    scene.animations = animations.data();
    scene.keyframes = keyframes.data();
}

namespace
{

// 1 + index of the object's animation in out.animations, 0 when the object is static
int32_t animationOf(const LoadedScene& loaded, const std::string& name, FlatScene& out,
                    std::map<std::string, int32_t>& seen)
{
    const auto it = loaded.animation.find(name);
    if (it == loaded.animation.end())
    {
        return 0;
    }
    const auto known = seen.find(name);
    if (known != seen.end())
    {
        return known->second;
    }
    const AnimatedObject& a = it->second;
    rt_animation ra{};
    ra.has_rotation = a.hasRotation ? 1 : 0;
    ra.rotation_axis[0] = a.rotationAxis.x;
    ra.rotation_axis[1] = a.rotationAxis.y;
    ra.rotation_axis[2] = a.rotationAxis.z;
    ra.base_rotation[0] = a.baseRotation.x;
    ra.base_rotation[1] = a.baseRotation.y;
    ra.base_rotation[2] = a.baseRotation.z;
    ra.base_rotation[3] = a.baseRotation.w;
    ra.constant_position[0] = a.constantPosition.x;
    ra.constant_position[1] = a.constantPosition.y;
    ra.constant_position[2] = a.constantPosition.z;
    ra.first_position_key = static_cast<int64_t>(out.keyframes.size());
    ra.position_key_count = static_cast<int64_t>(a.position.size());
    for (const Keyframe<Vector>& k : a.position)
    {
        rt_keyframe rk{};
        rk.time = k.time;
        rk.value[0] = k.value.x; rk.value[1] = k.value.y; rk.value[2] = k.value.z;
        rk.in_tangent[0] = k.inTangent.x; rk.in_tangent[1] = k.inTangent.y; rk.in_tangent[2] = k.inTangent.z;
        rk.out_tangent[0] = k.outTangent.x; rk.out_tangent[1] = k.outTangent.y; rk.out_tangent[2] = k.outTangent.z;
        rk.use_auto_tangent = k.useAutoTangent ? 1 : 0;
        out.keyframes.push_back(rk);
    }
    ra.first_rotation_key = static_cast<int64_t>(out.keyframes.size());
    ra.rotation_key_count = static_cast<int64_t>(a.rotationAngle.size());
    for (const Keyframe<double>& k : a.rotationAngle)
    {
        rt_keyframe rk{};
        rk.time = k.time;
        rk.value[0] = k.value;
        rk.in_tangent[0] = k.inTangent;
        rk.out_tangent[0] = k.outTangent;
        rk.use_auto_tangent = k.useAutoTangent ? 1 : 0;
        out.keyframes.push_back(rk);
    }
    out.animations.push_back(ra);
    const int32_t id = static_cast<int32_t>(out.animations.size());
    seen[name] = id;
    return id;
}

}  // namespace

void flattenScene(const LoadedScene& loaded, FlatScene& out)
{
    const RenderSettings& s = loaded.settings;
    rt_settings& rs = out.scene.settings;
    std::memset(&rs, 0, sizeof(rs));
    rs.photons_per_light = s.photonsPerLight;
    rs.bounce_store_capacity = s.bounceStoreCapacity;
    rs.fetch_size = s.fetchSize;
    rs.worker_count = s.workerCount;
    rs.probe_sub_sample = s.probeSubSample;
    rs.image_width = static_cast<int32_t>(s.imageWidth);
    rs.image_height = static_cast<int32_t>(s.imageHeight);
    rs.bounce_threshold = static_cast<int32_t>(s.bounceThreshold);
    rs.use_probe_gather = s.useProbeGather ? 1 : 0;
    rs.deterministic = s.deterministic ? 1 : 0;
    rs.seeded = (s.seed != RenderSettings::kUnseeded || s.deterministic) ? 1 : 0;
    rs.seed = (s.seed != RenderSettings::kUnseeded) ? s.seed : 0u;
    rs.camera_time_samples = s.cameraTimeSamples;
    rs.termination_threshold = s.terminationThreshold;
    rs.hash_grid_cell_size = s.hashGridCellSize;
    rs.density_cell_scale = s.densityCellScale;
    rs.splat_min_radius_scale = s.splatMinRadiusScale;
    rs.splat_luminance_clamp = s.splatLuminanceClamp;
    rs.probe_keep_radius_scale = s.probeKeepRadiusScale;
    rs.frame_time = s.frameTime;
    rs.shutter_time = s.shutterTime;

    // materials, library order (nine built-ins first)
    for (size_t i = 0; i < loaded.materialLibrary->size(); ++i)
    {
        const std::shared_ptr<Material> m = loaded.materialLibrary->fetchByIndex(i);
        // what the device runs for it: the material's own description (Material::deviceDescription)
        const DeviceMaterial d = m->deviceDescription();
        rt_material rm{};
        rm.type = static_cast<int32_t>(d.bsdf);
        rm.color[0] = d.color.red;
        rm.color[1] = d.color.green;
        rm.color[2] = d.color.blue;
        rm.alpha = d.alpha;
        rm.ior = d.ior;
        putName(rm.name, m->name());
        out.materials.push_back(rm);
    }

    // meshes, library order; volumes refer to them by index
    std::map<std::string, int32_t> animationIds;
    std::map<const Mesh*, int32_t> meshIndex;
    if (loaded.meshLibrary)
    {
        for (size_t i = 0; i < loaded.meshLibrary->size(); ++i)
        {
            const std::shared_ptr<Mesh> mesh = loaded.meshLibrary->fetchByIndex(i);
            rt_mesh rm{};
            rm.first_triangle = static_cast<int64_t>(out.triPos.size() / 9);
            rm.triangle_count = static_cast<int64_t>(mesh->triangles().size());
            putName(rm.name, mesh->name());
            for (const Triangle& t : mesh->triangles())
            {
                const Vector* p[3] = {&t.a, &t.b, &t.c};
                const Vector* n[3] = {&t.aNormal, &t.bNormal, &t.cNormal};
                for (int k = 0; k < 3; ++k)
                {
                    out.triPos.insert(out.triPos.end(), {p[k]->x, p[k]->y, p[k]->z});
                    out.triNrm.insert(out.triNrm.end(), {n[k]->x, n[k]->y, n[k]->z});
                }
            }
            meshIndex[mesh.get()] = static_cast<int32_t>(out.meshes.size());
            out.meshes.push_back(rm);
        }
    }

    for (const std::shared_ptr<Object>& object : loaded.objects)
    {
        if (auto volume = std::dynamic_pointer_cast<Volume>(object))
        {
            rt_volume rv{};
            rv.material = static_cast<int32_t>(volume->materialIndex());
            rv.mesh = -1;
            rv.animation = animationOf(loaded, object->name(), out, animationIds);
            put3(rv.position, volume->position());
            put4(rv.rotation, volume->rotation());
            if (auto mesh = std::dynamic_pointer_cast<MeshVolume>(object))
            {
                rv.type = RT_VOLUME_MESH;
                if (!mesh->mesh())
                {
                    throw std::runtime_error("MeshVolume \"" + object->name() + "\" has no mesh");
                }
                rv.mesh = meshIndex.at(mesh->mesh().get());
            }
            else if (auto sphere = std::dynamic_pointer_cast<SphereVolume>(object))
            {
                rv.type = RT_VOLUME_SPHERE;
                put3(rv.param, sphere->center());
                rv.param[3] = sphere->radius();
            }
            else if (auto quad = std::dynamic_pointer_cast<QuadVolume>(object))
            {
                rv.type = RT_VOLUME_QUAD;
                const Quad& q = quad->quad();
                put3(rv.param + 0, q.origin);
                put3(rv.param + 3, q.edgeU);
                put3(rv.param + 6, q.edgeV);
                put3(rv.param + 9, q.normal);
                put3(rv.param + 12, q.invU);
                put3(rv.param + 15, q.invV);
            }
            else if (auto plane = std::dynamic_pointer_cast<PlaneVolume>(object))
            {
                rv.type = RT_VOLUME_PLANE;
                put3(rv.param, plane->normal());
                rv.param[3] = plane->planeDot();
            }
            else
            {
                continue;  // bare Volume: casts nothing (reference src/Volume.cpp:51-55)
            }
            out.volumes.push_back(rv);
        }
        else if (auto light = std::dynamic_pointer_cast<Light>(object))
        {
            rt_light rl{};
            put3(rl.position, light->position());
            put4(rl.rotation, light->rotation());
            rl.color[0] = light->color().red;
            rl.color[1] = light->color().green;
            rl.color[2] = light->color().blue;
            rl.luminous_flux = light->luminousFlux();
            rl.intensity_candela = light->intensityCandela();
            rl.linearity = 1.0;
            putName(rl.name, light->name());
            if (auto omni = std::dynamic_pointer_cast<OmniLight>(object))
            {
                rl.type = RT_LIGHT_OMNI;
                rl.inner_radius = omni->innerRadius();
            }
            else if (auto spot = std::dynamic_pointer_cast<SpotLight>(object))
            {
                rl.type = RT_LIGHT_SPOT;
                rl.inner_radius = spot->innerRadius();
                rl.angle = spot->angle();
                rl.max_angle = spot->maxAngle();
            }
            else if (auto parallel = std::dynamic_pointer_cast<ParallelLight>(object))
            {
                rl.type = RT_LIGHT_PARALLEL;
                rl.radius = parallel->radius();
            }
            else if (auto area = std::dynamic_pointer_cast<AreaLight>(object))
            {
                rl.type = RT_LIGHT_AREA;
                rl.shape = area->shape() == AreaLight::Shape::Disc ? 1 : 0;
                rl.radius = area->radius();
                rl.width = area->width();
                rl.height = area->height();
                rl.flux_override = area->luminousFluxOverride();
            }
            else
            {
                continue;  // bare Light emits nothing (reference src/Light.cpp:48-51)
            }
            out.lights.push_back(rl);
        }
    }

    for (const std::shared_ptr<Camera>& camera : loaded.cameras)
    {
        rt_camera rc{};
        rc.animation = animationOf(loaded, camera->name(), out, animationIds);
        put3(rc.position, camera->position());
        put4(rc.rotation, camera->rotation());
        rc.vertical_fov_degrees = camera->verticalFieldOfView();
        rc.width = static_cast<int32_t>(camera->width());
        rc.height = static_cast<int32_t>(camera->height());
        rc.projection = static_cast<int32_t>(camera->projection());
        rc.bounce_filter = camera->bounceFilter();
        rc.light_filter = camera->lightFilter();
        rc.orthographic_height = camera->orthographicHeight();
        rc.aperture_radius = camera->effectiveApertureRadius();
        rc.focus_distance = camera->focusDistance();
        rc.f_number = camera->fNumber();
        rc.shutter_time = camera->shutterTime();
        rc.iso = camera->iso();
        putName(rc.name, camera->name());
        out.cameras.push_back(rc);
    }
    out.bind();
}
