// Wavefront OBJ reader + MeshLibrary::addFromFile (reference src/ObjReader.cpp:15-96,
// src/MeshLibrary.cpp:13-26).
//
// The reference loads OBJ files through tinyobjloader: one Mesh per `o`/`g` shape,
// polygons triangulated as a fan (v0,v1,v2),(v0,v2,v3)... (tests/test_TriangleWatertight
// .cpp:15-17 documents the order), vertex data parsed to FLOAT and widened to double,
// every face vertex must carry a normal.  This reader follows those rules.
//
// The reference's own meshes/*.obj are Git-LFS pointer files whose payload is not
// available offline.  When a scene names one of them (or the file is missing) a
// procedural stand-in with the same shape names is generated instead
// (MeshGenerator.cpp); any other unreadable file is an error.

#include "ObjReader.h"

#include "MeshGenerator.h"

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

namespace
{

bool isLfsPointer(const std::filesystem::path& path)
{
    std::ifstream file(path);
    std::string line;
    return file && std::getline(file, line) && line.rfind("version https://git-lfs", 0) == 0;
}

double asFloat(const std::string& token)
{
    return static_cast<double>(static_cast<float>(std::strtod(token.c_str(), nullptr)));
}

struct FaceVertex
{
    long long v = 0, vn = 0;
};

FaceVertex parseFaceVertex(const std::string& token)
{
    // forms: v, v/vt, v//vn, v/vt/vn
    FaceVertex fv;
    const size_t s1 = token.find('/');
    fv.v = std::strtoll(token.substr(0, s1).c_str(), nullptr, 10);
    if (s1 != std::string::npos)
    {
        const size_t s2 = token.find('/', s1 + 1);
        if (s2 != std::string::npos && s2 + 1 < token.size())
        {
            fv.vn = std::strtoll(token.substr(s2 + 1).c_str(), nullptr, 10);
        }
    }
    return fv;
}

}  // namespace

namespace ObjReader
{

std::vector<std::shared_ptr<Mesh>> loadMeshes(const std::filesystem::path& path, bool logToStdout)
{
    if (!std::filesystem::exists(path) || isLfsPointer(path))
    {
        std::vector<std::shared_ptr<Mesh>> generated = MeshGenerator::standInFor(path.stem().string());
        if (generated.empty())
        {
            throw std::runtime_error("ObjReader::loadMeshes error: cannot read " + path.string() +
                                     " and no stand-in mesh is known by that name");
        }
        if (logToStdout)
        {
            std::cout << "Mesh file " << path.generic_string() << " unavailable; generated stand-in shapes" << std::endl;
        }
        return generated;
    }

    std::ifstream file(path);
    std::vector<Vector> positions, normals;
    std::vector<std::shared_ptr<Mesh>> meshes;
    std::vector<Triangle> triangles;
    std::string shapeName;
    auto flush = [&]() {
        if (!triangles.empty())
        {
            meshes.push_back(std::make_shared<Mesh>(shapeName, triangles));
            triangles.clear();
        }
    };

    std::string line;
    while (std::getline(file, line))
    {
        std::istringstream in(line);
        std::string tag;
        in >> tag;
        if (tag == "v" || tag == "vn")
        {
            std::string x, y, z;
            in >> x >> y >> z;
            (tag == "v" ? positions : normals).push_back(Vector{asFloat(x), asFloat(y), asFloat(z)});
        }
        else if (tag == "o" || tag == "g")
        {
            flush();
            std::getline(in >> std::ws, shapeName);
            while (!shapeName.empty() && (shapeName.back() == '\r' || shapeName.back() == ' '))
            {
                shapeName.pop_back();
            }
        }
        else if (tag == "f")
        {
            std::vector<FaceVertex> face;
            std::string token;
            while (in >> token)
            {
                face.push_back(parseFaceVertex(token));
            }
            auto resolve = [](long long index, size_t count) -> size_t {
                const long long zeroBased = index > 0 ? index - 1 : static_cast<long long>(count) + index;
                if (index == 0 || zeroBased < 0 || static_cast<size_t>(zeroBased) >= count)
                {
                    throw std::runtime_error("ObjReader::loadMeshes error: face index out of range (every face vertex "
                                             "needs a position and a normal)");
                }
                return static_cast<size_t>(zeroBased);
            };
            for (size_t k = 1; k + 1 < face.size(); ++k)
            {
                const FaceVertex fv[3] = {face[0], face[k], face[k + 1]};
                Triangle t;
                t.a = positions[resolve(fv[0].v, positions.size())];
                t.b = positions[resolve(fv[1].v, positions.size())];
                t.c = positions[resolve(fv[2].v, positions.size())];
                t.aNormal = normals[resolve(fv[0].vn, normals.size())];
                t.bNormal = normals[resolve(fv[1].vn, normals.size())];
                t.cNormal = normals[resolve(fv[2].vn, normals.size())];
                triangles.push_back(t);
            }
        }
    }
    flush();
    if (logToStdout)
    {
        std::cout << "Loaded " << meshes.size() << " meshes from " << path.generic_string() << std::endl;
    }
    return meshes;
}

}  // namespace ObjReader

void MeshLibrary::addFromFile(const std::filesystem::path& path)
{
    std::vector<std::shared_ptr<Mesh>> meshes = ObjReader::loadMeshes(path);
    for (auto& mesh : meshes)
    {
        if (mesh->name().empty())
        {
            mesh->name("mesh_" + std::to_string(size()));
        }
        add(mesh);
    }
}
