// ScenePresets.h — the Cornell-box scene script of the reference application
// (reference source/Main.cpp:26-151,165-208; constants in SURVEY.md App. D) as a
// reusable function, plus the build-defined variants BASELINE.json's configs
// name: the mesh in "slot 3" is swapped, everything else stays.
#pragma once

#include <string>

#include "Scene.h"

namespace rtpreset {

// kind: "cubes"  slot 3 = cube_tri.off             (the shipped program, C1/C3)
//       "lowres" slot 3 = example_low_res.off      (C2, 1,222 triangles)
//       "hires"  slot 3 = example.off              (C4, 11,666 triangles)
//       "stress" slot 3 = 83,334 lattice copies of cube_tri.off (C5, 1,000,008 tris)
//       "file:<name.off>" slot 3 = that file from meshDir
// Throws std::runtime_error on an unknown kind or unreadable mesh.
Scene buildCornellScene(const std::string& kind, const std::string& meshDir, size_t width, size_t height);

// Main.cpp:88-99: rotate POSITIONS about +Y; normals are left as loaded.
void rotationY(Mesh& mesh, float phi);

}  // namespace rtpreset
