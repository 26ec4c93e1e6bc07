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

BSDFSample Material::sample(const Vector& incident, const Vector& normal, double u1, double u2) const
{
    const double row[14] = {static_cast<double>(static_cast<int>(m_kind)), m_color.red, m_color.green, m_color.blue,
                            m_alpha, m_ior, incident.x, incident.y, incident.z, normal.x, normal.y, normal.z, u1, u2};
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
    const double row[14] = {static_cast<double>(static_cast<int>(m_kind)), m_color.red, m_color.green, m_color.blue,
                            m_alpha, wi.x, wi.y, wi.z, wo.x, wo.y, wo.z, normal.x, normal.y, normal.z};
    double out[4] = {};
    check(rt_bsdf_evaluate(0, 1, row, out));
    return Color{static_cast<float>(out[0]), static_cast<float>(out[1]), static_cast<float>(out[2])};
}

double Material::pdf(const Vector& wi, const Vector& wo, const Vector& normal) const
{
    const double row[14] = {static_cast<double>(static_cast<int>(m_kind)), m_color.red, m_color.green, m_color.blue,
                            m_alpha, wi.x, wi.y, wi.z, wo.x, wo.y, wo.z, normal.x, normal.y, normal.z};
    double out[4] = {};
    check(rt_bsdf_evaluate(0, 1, row, out));
    return out[3];
}
