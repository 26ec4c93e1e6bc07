// ray-tracer <scene.json> — the command line of the reference (src/main.cpp:22-154) in front of
// the device renderer.  What a caller can observe is kept: the greeting and farewell lines, the
// usage messages, "ERROR: ..." with exit status 1 for anything that throws, one
// <renderPath>/<renderName>.<frame>.png per frame (plus one per further camera that names an
// output), and the per-frame report lines.  The structure is this program's own: a small
// command-line front (usage checks), a frame schedule, and one report object per frame.
#include "PngWriter.h"
#include "Renderer.h"
#include "SceneLoader.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <optional>
#include <string>
#include <vector>

namespace
{

namespace fs = std::filesystem;
using Clock = std::chrono::steady_clock;

void say(const std::string& line) { std::cout << line << std::endl; }

// The project file named on the command line, or nothing after the matching usage message.
std::optional<fs::path> projectFileFrom(const std::vector<std::string>& arguments)
{
    if (arguments.empty())
    {
        say("Please provide a project file to process");
        return std::nullopt;
    }
    if (arguments.size() != 1)
    {
        say("Too many arguments, only 1 required");
        return std::nullopt;
    }
    const fs::path candidate = arguments.front();
    if (!fs::is_regular_file(candidate))
    {
        std::cout << "Provided path is invalid: " << candidate << std::endl;
        return std::nullopt;
    }
    return candidate;
}

// The instant a frame's shutter opens: frameOffset + frame / frameRate (24 frames a second when
// the scene names no rate).  This is the one place a frame number becomes a scene time.
double shutterOpens(const RenderSettings& settings, size_t frame)
{
    const double perSecond = settings.frameRate > 0.0 ? settings.frameRate : 24.0;
    return settings.frameOffset + static_cast<double>(frame) / perSecond;
}

// Raw float buffers next to the PNGs (RT_DUMP_BUFFERS=<dir>): camera<c>.f32 holds
// height x width x 3 floats, for comparisons that the 16-bit quantisation would blur.
void dumpBuffers(const RenderResult& render, const fs::path& directory)
{
    fs::create_directories(directory);
    size_t index = 0;
    for (const CameraRender& camera : render.cameras)
    {
        const Buffer& buffer = *camera.buffer;
        std::ofstream out(directory / ("camera" + std::to_string(index++) + ".f32"), std::ios::binary);
        out.write(reinterpret_cast<const char*>(buffer.data()),
                  static_cast<std::streamsize>(buffer.width() * buffer.height() * 3 * sizeof(float)));
    }
}

void writePng(const fs::path& path, const Image& image, const std::string& title)
{
    if (!PngWriter::writeImage(path, image, title))
    {
        throw std::runtime_error("cannot write " + path.generic_string());
    }
    say("Wrote " + path.generic_string());
}

// Everything printed about one finished frame.
struct FrameReport
{
    const LoadedScene& scene;
    const RenderResult& render;

    void photonPass() const
    {
        std::cout << "Photon pass: photons=" << render.photons << " casts=" << render.casts << " bounces=" << render.bounces
                  << " seconds=" << render.photonPassSeconds << std::endl;
    }

    void probeGather() const
    {
        if (!scene.settings.useProbeGather)
        {
            return;
        }
        const size_t tested = render.bounceKept + render.bounceCulled;
        const double culledShare = tested == 0 ? 0.0 : 100.0 * static_cast<double>(render.bounceCulled) / static_cast<double>(tested);
        const bool overflowed = render.bounceStoreDropped > 0;
        std::cout << "Probe gather: kept=" << render.bounceKept << " culled=" << render.bounceCulled << " (" << culledShare
                  << "% culled) storeSize=" << render.bounceStoreSize
                  << " storeMiB=" << (render.bounceStoreCapacity * 52 / (1024 * 1024)) << (overflowed ? " [BUDGET HIT]" : "")
                  << std::endl;
        if (overflowed)
        {
            std::cout << "Probe gather: DROPPED " << render.bounceStoreDropped
                      << " deposits (BounceStore overflow — image is missing energy; raise $bounceStoreCapacity)" << std::endl;
        }
    }

    void timing(std::chrono::microseconds spent) const
    {
        say("Render time:");
        std::cout << "|- total:        " << spent.count() / 1000 << " ms" << std::endl;
        const size_t pixels = scene.settings.imageWidth * scene.settings.imageHeight;
        if (pixels > 0)
        {
            std::cout << "|- average / px: " << spent.count() / pixels << " us" << std::endl;
        }
    }
};

void renderOneFrame(LoadedScene& scene, size_t frame, size_t ordinal, size_t frameTotal)
{
    say("---");
    std::cout << "Rendering frame " << ordinal << " / " << frameTotal << std::endl;
    const Clock::time_point started = Clock::now();
    scene.settings.frameTime = shutterOpens(scene.settings, frame);
    std::cout << "Frame time t=" << scene.settings.frameTime << "s shutter=" << scene.settings.shutterTime << "s" << std::endl;

    const RenderResult render = Renderer::renderFrame(scene);
    const auto spent = std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - started);

    const FrameReport report{scene, render};
    report.photonPass();
    report.probeGather();

    const std::string suffix = "." + std::to_string(frame) + ".png";
    const Clock::time_point writing = Clock::now();
    writePng(scene.renderPath / (scene.renderName + suffix), *render.image, scene.renderName);
    if (std::getenv("RT_AMD_VERBOSE"))
    {
        std::fprintf(stderr, "[rt_amd] %-28s %8.1f ms\n", "PNG written",
                     std::chrono::duration<double, std::milli>(Clock::now() - writing).count());
    }
    for (size_t c = 1; c < render.cameras.size(); ++c)
    {
        const CameraRender& extra = render.cameras[c];
        if (!extra.outputName.empty() && extra.image)
        {
            writePng(scene.renderPath / (extra.outputName + suffix), *extra.image, extra.outputName);
        }
    }
    if (const char* directory = std::getenv("RT_DUMP_BUFFERS"))
    {
        dumpBuffers(render, directory);
    }
    report.timing(spent);
}

int renderProject(const fs::path& projectFile)
{
    LoadedScene scene = SceneLoader::loadFromFile(projectFile, /*logToStdout=*/true);
    const size_t first = scene.settings.startFrame, last = scene.settings.endFrame;
    say("---");
    std::cout << "Rendering image at " << scene.settings.imageWidth << " px by " << scene.settings.imageHeight << " px"
              << std::endl;
    const size_t frameTotal = last - first + 1;
    for (size_t frame = first; frame <= last; ++frame)
    {
        renderOneFrame(scene, frame, frame + 1, frameTotal);
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    say("Hello!");
    const std::optional<fs::path> projectFile = projectFileFrom(std::vector<std::string>(argv + (argc > 0 ? 1 : 0), argv + argc));
    if (!projectFile)
    {
        return 1;
    }
    try
    {
        renderProject(*projectFile);
    }
    catch (const std::exception& failure)
    {
        // a failed render has to fail the process: scripts and CI gates read the exit status
        say(std::string("ERROR: ") + failure.what());
        return 1;
    }
    say("Goodbye!");
    return 0;
}
