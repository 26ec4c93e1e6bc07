// C entry points over the C++ host layer (include/rt_host.h).

#include "rt_host.h"

#include "Flatten.h"
#include "FrameSetup.h"
#include "PngWriter.h"
#include "RankFrames.h"
#include "SceneLoader.h"
#include "rt_scene_io.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>

struct rth_scene
{
    LoadedScene loaded;
    FlatScene flat;
};

namespace
{
thread_local std::string g_error;
}

extern "C" {

const char* rth_last_error(void) { return g_error.c_str(); }

int rth_scene_load(const char* json_path, rth_scene** out)
{
    *out = nullptr;
    try
    {
        auto scene = std::make_unique<rth_scene>();
        scene->loaded = SceneLoader::loadFromFile(json_path);
        flattenScene(scene->loaded, scene->flat);
        *out = scene.release();
        return 0;
    }
    catch (const std::exception& e)
    {
        g_error = e.what();
        return 1;
    }
}

const rt_scene* rth_scene_get(const rth_scene* scene) { return &scene->flat.scene; }

rt_settings* rth_scene_settings(rth_scene* scene) { return &scene->flat.scene.settings; }

int rth_scene_write(const rth_scene* scene, const char* path)
{
    const int rc = rt_scene_write(&scene->flat.scene, path);
    if (rc != 0)
    {
        g_error = std::string("cannot write ") + path;
    }
    return rc;
}

void rth_scene_free(rth_scene* scene) { delete scene; }

int rth_frame_derive(const rt_scene* scene, double out[4])
{
    const FrameDerived d = deriveFrame(*scene);
    out[0] = d.sceneDepthFootprint;
    out[1] = d.cellSize;
    out[2] = d.minSplatRadius;
    out[3] = d.keepRadius;
    return 0;
}

int rth_material_sample(const rth_scene* scene, int32_t material, const double in6[6], uint32_t seed, double out9[9])
{
    try
    {
        const std::shared_ptr<Material> m = scene->loaded.materialLibrary->fetchByIndex(static_cast<size_t>(material));
        RandomGenerator generator(seed), replay(seed);
        const BSDFSample sample = m->sample(Vector{in6[0], in6[1], in6[2]}, Vector{in6[3], in6[4], in6[5]}, generator);
        out9[0] = sample.valid ? 1.0 : 0.0;
        out9[1] = sample.direction.x;
        out9[2] = sample.direction.y;
        out9[3] = sample.direction.z;
        out9[4] = sample.weight.red;
        out9[5] = sample.weight.green;
        out9[6] = sample.weight.blue;
        out9[7] = sample.pdf;
        // draws consumed: the first value the two generators disagree on
        // draws consumed: how far a generator of the same seed has to run to meet this one's next value
        int draws = 0;
        const double next = generator.value();
        for (; draws < 8; ++draws)
        {
            if (replay.value() == next) break;
        }
        out9[8] = static_cast<double>(draws);
        return 0;
    }
    catch (const std::exception& e)
    {
        g_error = e.what();
        return 1;
    }
}

int rth_material_evaluate(const rth_scene* scene, int32_t material, const double in9[9], double out5[5])
{
    try
    {
        const std::shared_ptr<Material> m = scene->loaded.materialLibrary->fetchByIndex(static_cast<size_t>(material));
        const Vector wi{in9[0], in9[1], in9[2]}, wo{in9[3], in9[4], in9[5]}, normal{in9[6], in9[7], in9[8]};
        const Color f = m->evaluate(wi, wo, normal);
        out5[0] = f.red;
        out5[1] = f.green;
        out5[2] = f.blue;
        out5[3] = m->pdf(wi, wo, normal);
        out5[4] = m->isDelta() ? 1.0 : 0.0;
        return 0;
    }
    catch (const std::exception& e)
    {
        g_error = e.what();
        return 1;
    }
}

int rth_write_png16(const char* path, const uint16_t* rgb, int32_t width, int32_t height, const char* title)
{
    if (path == nullptr || rgb == nullptr || width <= 0 || height <= 0)
    {
        g_error = "rth_write_png16: bad arguments";
        return 1;
    }
    Image image(static_cast<size_t>(width), static_cast<size_t>(height));
    for (int32_t y = 0; y < height; ++y)
    {
        for (int32_t x = 0; x < width; ++x)
        {
            const uint16_t* p = rgb + 3 * (static_cast<size_t>(y) * static_cast<size_t>(width) + static_cast<size_t>(x));
            image.setPixel(static_cast<size_t>(x), static_cast<size_t>(y), Pixel{p[0], p[1], p[2]});
        }
    }
    if (!PngWriter::writeImage(path, image, title ? title : ""))
    {
        g_error = std::string("cannot write ") + path;
        return 1;
    }
    return 0;
}

int rth_test_rank_frames(int32_t world, int32_t failing_rank, int32_t cancel_after_pieces, int32_t* entered_collective,
                         int32_t* pieces_traced, char* text, int32_t text_capacity)
{
    if (world < 1 || world > 64 || entered_collective == nullptr || pieces_traced == nullptr || text == nullptr || text_capacity < 1)
    {
        g_error = "rth_test_rank_frames: bad arguments";
        return 1;
    }
    constexpr int kPieces = 8;
    RankFrames frames(world);
    std::atomic<int> entered{0}, traced{0};
    bool stuck = false;
    const std::string failure = frames.run(
        [&](int rank) {
            // a share: eight pieces of photons, none started once the frame is being given up
            for (int piece = 0; piece < kPieces && !frames.abandoned(); ++piece)
            {
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
                traced.fetch_add(1);
                if (rank == 0 && cancel_after_pieces >= 0 && piece + 1 >= cancel_after_pieces)
                {
                    frames.cancel();
                }
            }
            if (rank == failing_rank)
            {
                throw std::runtime_error("injected failure of rank " + std::to_string(rank));
            }
        },
        [&](int) {
            // a collective: nobody leaves before everybody has entered (given up after two seconds, and
            // reported, so that a broken meeting fails the test instead of hanging it)
            entered.fetch_add(1);
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
            while (entered.load() < world)
            {
                if (std::chrono::steady_clock::now() > deadline)
                {
                    stuck = true;
                    throw std::runtime_error("collective entered by " + std::to_string(entered.load()) + " of " +
                                             std::to_string(world) + " ranks: it would hang");
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
        });
    *entered_collective = entered.load();
    *pieces_traced = traced.load();
    std::strncpy(text, failure.c_str(), static_cast<size_t>(text_capacity) - 1);
    text[text_capacity - 1] = '\0';
    return stuck ? 2 : 0;
}

}  // extern "C"
