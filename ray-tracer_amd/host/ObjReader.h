// OBJ mesh loading (reference include/ObjReader.h).
#pragma once

#include "Scene.h"

#include <filesystem>
#include <memory>
#include <vector>

namespace ObjReader
{

std::vector<std::shared_ptr<Mesh>> loadMeshes(const std::filesystem::path& path, bool logToStdout = false);

}
