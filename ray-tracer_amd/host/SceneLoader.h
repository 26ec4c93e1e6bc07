// Scene-file loader with the reference's entry point and key set
// (reference include/SceneLoader.h:60, src/SceneLoader.cpp:765-1033).
#pragma once

#include "Scene.h"

#include <filesystem>

namespace SceneLoader
{

// Throws std::runtime_error (or Json::ParseError, a runtime_error) on bad input,
// as the reference throws runtime_error / nlohmann parse exceptions.
LoadedScene loadFromFile(const std::filesystem::path& scenePath, bool logToStdout = false);

}
