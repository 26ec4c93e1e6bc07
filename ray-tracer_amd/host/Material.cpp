// Material::sample / evaluate / pdf of the host classes (reference include/Material.h:34-128),
// evaluated by the device library's function-level entry points so that the host-side
// interface and the photon pass share one implementation of every BSDF.
#include "Scene.h"

#include "rt_amd.h"

#include <stdexcept>
#include <string>

namespace
{
void check(int32_t code)
{
    if (code != RT_OK)
    {
        throw std::runtime_error(std::string("rt_amd: ") + rt_last_error());
    }
}
}  // namespace

BSDFSample Material::sample(const Vector& incident, const Vector& normal, RandomGenerator& generator) const
{
    double u1 = 0.0, u2 = 0.0;
    switch (deviceDescription().bsdf)
    {
        case RT_MATERIAL_LAMBERTIAN:
            u1 = generator.value();
            u2 = generator.value();
            break;
        case RT_MATERIAL_DIELECTRIC:
            u1 = generator.value();
            break;
        case RT_MATERIAL_MICROFACET:
        {
            // no draw for light that arrives from behind (reference src/MicrofacetMaterial.cpp:144-153)
            const double cosThetaI = (0.0 - incident.x) * normal.x + (0.0 - incident.y) * normal.y + (0.0 - incident.z) * normal.z;
            if (cosThetaI > 0.0)
            {
                u1 = generator.value();
                u2 = generator.value();
            }
            break;
        }
        default:
            break;  // a mirror draws nothing
    }
    return sample(incident, normal, u1, u2);
}

BSDFSample Material::sample(const Vector& incident, const Vector& normal, double u1, double u2) const
{
    const DeviceMaterial d = deviceDescription();
    const double row[14] = {static_cast<double>(d.bsdf), d.color.red, d.color.green, d.color.blue,
                            d.alpha, d.ior, incident.x, incident.y, incident.z, normal.x, normal.y, normal.z, u1, u2};
    double out[10] = {};
    check(rt_bsdf_sample(0, 1, row, out));
    BSDFSample s;
    s.valid = out[0] != 0.0;
    s.direction = Vector{out[1], out[2], out[3]};
    s.weight = Color{static_cast<float>(out[4]), static_cast<float>(out[5]), static_cast<float>(out[6])};
    s.pdf = out[7];
    s.isDelta = out[8] != 0.0;
    return s;
}

Color Material::evaluate(const Vector& wi, const Vector& wo, const Vector& normal) const
{
    const DeviceMaterial d = deviceDescription();
    const double row[14] = {static_cast<double>(d.bsdf), d.color.red, d.color.green, d.color.blue,
                            d.alpha, wi.x, wi.y, wi.z, wo.x, wo.y, wo.z, normal.x, normal.y, normal.z};
    double out[4] = {};
    check(rt_bsdf_evaluate(0, 1, row, out));
    return Color{static_cast<float>(out[0]), static_cast<float>(out[1]), static_cast<float>(out[2])};
}

double Material::pdf(const Vector& wi, const Vector& wo, const Vector& normal) const
{
    const DeviceMaterial d = deviceDescription();
    const double row[14] = {static_cast<double>(d.bsdf), d.color.red, d.color.green, d.color.blue,
                            d.alpha, wi.x, wi.y, wi.z, wo.x, wo.y, wo.z, normal.x, normal.y, normal.z};
    double out[4] = {};
    check(rt_bsdf_evaluate(0, 1, row, out));
    return out[3];
}
