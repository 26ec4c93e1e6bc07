// Renderer: one frame of a loaded scene on one MI355X, through the C-ABI of
// include/rt_amd.h.  Mirrors the reference's interface (include/Renderer.h:21-144):
// renderFrame returns the per-camera float buffers and 16-bit images plus the pass
// diagnostics; tonemapBufferToImage is the reference's exposure + gamma.
#pragma once

#include "Scene.h"

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

// Linear float accumulation buffer, [height][width][3] (reference include/Buffer.h).
class Buffer
{
public:
    Buffer(size_t width, size_t height) : m_width(width), m_height(height), m_data(width * height * 3, 0.0f) {}
    size_t width() const { return m_width; }
    size_t height() const { return m_height; }
    Color fetchColor(size_t x, size_t y) const
    {
        const size_t at = (y * m_width + x) * 3;
        return Color{m_data[at], m_data[at + 1], m_data[at + 2]};
    }
    float* data() { return m_data.data(); }
    const float* data() const { return m_data.data(); }

private:
    size_t m_width, m_height;
    std::vector<float> m_data;
};

struct Pixel
{
    std::uint16_t red = 0, green = 0, blue = 0;
};

// 16-bit RGB image (reference include/Image.h).
class Image
{
public:
    Image(size_t width, size_t height) : m_width(width), m_height(height), m_pixels(width * height) {}
    size_t width() const { return m_width; }
    size_t height() const { return m_height; }
    void setPixel(size_t x, size_t y, Pixel p) { m_pixels[y * m_width + x] = p; }
    const Pixel& getPixel(size_t x, size_t y) const { return m_pixels[y * m_width + x]; }
    void clear() { m_pixels.assign(m_pixels.size(), Pixel{}); }

private:
    size_t m_width, m_height;
    std::vector<Pixel> m_pixels;
};

// Gather diagnostics of one camera (reference include/ProbeGather.h:106-125).
struct ProbeGatherResult
{
    size_t pixelsHit = 0, pixelsGathered = 0, depositsAccum = 0;
    size_t cameraRays = 0, deltaExtensions = 0, misses = 0;
};

struct CameraRender  // reference include/Renderer.h:21-49
{
    std::shared_ptr<Camera> camera;
    std::string outputName;
    std::shared_ptr<Buffer> buffer;
    std::shared_ptr<Image> image;
    ProbeGatherResult probe;
    double gatherSeconds = 0.0;
    double meanLuminance = 0.0;
};

struct RenderResult  // reference include/Renderer.h:51-103
{
    std::shared_ptr<Buffer> buffer;  // primary camera
    std::shared_ptr<Image> image;
    std::vector<CameraRender> cameras;
    double photonPassSeconds = 0.0;
    size_t bounceStoreSize = 0, bounceStoreCapacity = 0, bounceStoreAttempted = 0, bounceStoreDropped = 0;
    size_t emitterDepositsKept = 0;
    size_t bounceKept = 0, bounceCulled = 0;
    size_t photons = 0, casts = 0, bounces = 0;
    size_t splatTotal = 0, splatRadiusClamped = 0, splatLuminanceClamped = 0;
    size_t probeCount = 0;
};

namespace Renderer
{

// remainingWork -> keep going?  Called from the calling thread while the pass runs.
using ProgressCallback = std::function<bool(size_t remainingWork)>;
using PreviewCallback = std::function<void(const Buffer&, double emittedFraction, double saturationLuminance)>;

// GPU the frame runs on (default 0).
void setDevice(int device);

// reference src/Renderer.cpp:40-76
void tonemapBufferToImage(const Buffer& buffer, Image& image, double photonsEmitted, double saturationLuminance);

// reference src/Renderer.cpp:78-809.  Throws std::runtime_error on any failure of the
// device library.
RenderResult renderFrame(const LoadedScene& scene, ProgressCallback progress = nullptr,
                         PreviewCallback preview = nullptr);

}  // namespace Renderer
