// ray-tracer <scene.json>: the reference's command line (src/main.cpp:22-154).  Loads the
// scene, renders frames startFrame..endFrame on GPU 0, writes <renderPath>/<renderName>.<frame>.png
// (and one PNG per additional camera that names an output), exits 1 on any error.
#include "PngWriter.h"
#include "Renderer.h"
#include "SceneLoader.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>

namespace
{

// Optional raw dump of the float buffers (RT_DUMP_BUFFERS=<dir>): camera<c>.f32 holds
// height x width x 3 floats; used to compare frames without the 16-bit quantisation.
void dumpBuffers(const RenderResult& render, const std::filesystem::path& dir)
{
    std::filesystem::create_directories(dir);
    for (size_t c = 0; c < render.cameras.size(); ++c)
    {
        const Buffer& buffer = *render.cameras[c].buffer;
        std::ofstream out(dir / ("camera" + std::to_string(c) + ".f32"), std::ios::binary);
        out.write(reinterpret_cast<const char*>(buffer.data()),
                  static_cast<std::streamsize>(buffer.width() * buffer.height() * 3 * sizeof(float)));
    }
}

}  // namespace

int main(int argc, char** argv)
{
    std::cout << "Hello!" << std::endl;
    if (argc == 1)
    {
        std::cout << "Please provide a project file to process" << std::endl;
        return 1;
    }
    else if (argc > 2)
    {
        std::cout << "Too many arguments, only 1 required" << std::endl;
        return 1;
    }
    const std::filesystem::path projectFilePath = argv[1];
    if (!std::filesystem::is_regular_file(projectFilePath))
    {
        std::cout << "Provided path is invalid: " << projectFilePath << std::endl;
        return 1;
    }
    try
    {
        LoadedScene scene = SceneLoader::loadFromFile(projectFilePath, /*logToStdout=*/true);
        const size_t startFrame = scene.settings.startFrame;
        const size_t endFrame = scene.settings.endFrame;
        const size_t frameCount = (endFrame - startFrame) + 1;
        const size_t pixelCount = scene.settings.imageWidth * scene.settings.imageHeight;
        std::cout << "---" << std::endl;
        std::cout << "Rendering image at " << scene.settings.imageWidth << " px by " << scene.settings.imageHeight
                  << " px" << std::endl;
        for (size_t frame = startFrame; frame <= endFrame; ++frame)
        {
            std::cout << "---" << std::endl;
            std::cout << "Rendering frame " << frame + 1 << " / " << frameCount << std::endl;
            const auto renderStart = std::chrono::steady_clock::now();
            const double frameRate = (scene.settings.frameRate > 0.0) ? scene.settings.frameRate : 24.0;
            scene.settings.frameTime = scene.settings.frameOffset + static_cast<double>(frame) / frameRate;
            std::cout << "Frame time t=" << scene.settings.frameTime << "s"
                      << " shutter=" << scene.settings.shutterTime << "s" << std::endl;

            const RenderResult render = Renderer::renderFrame(scene);

            const auto renderDuration = std::chrono::duration_cast<std::chrono::microseconds>(
                std::chrono::steady_clock::now() - renderStart);
            std::cout << "Photon pass: photons=" << render.photons << " casts=" << render.casts
                      << " bounces=" << render.bounces << " seconds=" << render.photonPassSeconds << std::endl;
            if (scene.settings.useProbeGather)
            {
                const size_t total = render.bounceKept + render.bounceCulled;
                const double cullPct =
                    total > 0 ? 100.0 * static_cast<double>(render.bounceCulled) / static_cast<double>(total) : 0.0;
                std::cout << "Probe gather: kept=" << render.bounceKept << " culled=" << render.bounceCulled << " ("
                          << cullPct << "% culled)"
                          << " storeSize=" << render.bounceStoreSize
                          << " storeMiB=" << (render.bounceStoreCapacity * 52 / (1024 * 1024))
                          << (render.bounceStoreDropped > 0 ? " [BUDGET HIT]" : "") << std::endl;
                if (render.bounceStoreDropped > 0)
                {
                    std::cout << "Probe gather: DROPPED " << render.bounceStoreDropped
                              << " deposits (BounceStore overflow — image is missing energy; raise "
                                 "$bounceStoreCapacity)"
                              << std::endl;
                }
            }
            const std::string fileName = scene.renderName + "." + std::to_string(frame) + ".png";
            const std::filesystem::path outputPath = scene.renderPath / fileName;
            const auto writeStart = std::chrono::steady_clock::now();
            if (!PngWriter::writeImage(outputPath, *render.image, scene.renderName))
            {
                throw std::runtime_error("cannot write " + outputPath.generic_string());
            }
            std::cout << "Wrote " << outputPath.generic_string() << std::endl;
            if (std::getenv("RT_AMD_VERBOSE"))
            {
                std::fprintf(stderr, "[rt_amd] %-28s %8.1f ms\n", "PNG written",
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - writeStart).count());
            }
            for (size_t c = 1; c < render.cameras.size(); ++c)
            {
                const CameraRender& cr = render.cameras[c];
                if (cr.outputName.empty() || !cr.image)
                {
                    continue;
                }
                const std::filesystem::path extra =
                    scene.renderPath / (cr.outputName + "." + std::to_string(frame) + ".png");
                if (!PngWriter::writeImage(extra, *cr.image, cr.outputName))
                {
                    throw std::runtime_error("cannot write " + extra.generic_string());
                }
                std::cout << "Wrote " << extra.generic_string() << std::endl;
            }
            if (const char* dump = std::getenv("RT_DUMP_BUFFERS"))
            {
                dumpBuffers(render, dump);
            }
            std::cout << "Render time:" << std::endl;
            std::cout << "|- total:        " << renderDuration.count() / 1000 << " ms" << std::endl;
            if (pixelCount > 0)
            {
                std::cout << "|- average / px: " << renderDuration.count() / pixelCount << " us" << std::endl;
            }
        }
    }
    catch (const std::exception& e)
    {
        std::cout << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "Goodbye!" << std::endl;
    return 0;
}
