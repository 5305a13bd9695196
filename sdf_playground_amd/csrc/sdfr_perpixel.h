// sdfr_perpixel.h -- the per-pixel pipeline (sdfr_render_pixel.h) together with every scene
// compiled ahead of time, and the registry that numbers them.
#pragma once
#include "sdfr_render_pixel.h"
#include "sdfr_scenes.h"
#include "sdfr_scenes2.h"
#include "sdfr_scenes3.h"
#include "sdfr_scenes4.h"
#include "sdfr_scene_debug.h"

namespace sdfr {

// scene registry: X(index, SceneType)
#define SDFR_FOR_EACH_SCENE(X) \
	X(0, SceneFastSphere) X(1, SceneCubeSea) X(2, SceneLabyrinth) X(3, SceneFractal) X(4, SceneLense) X(5, SceneGems) X(6, SceneLightShadows) \
	X(7, SceneCube) X(8, SceneGyroid) X(9, SceneBasicTransparency) X(10, SceneBasicClouds) X(11, SceneCoordinateMaterial) \
	X(12, SceneDistortion) X(13, SceneTable) X(14, SceneSierpinski) X(15, SceneNeon) \
	X(16, SceneFractal2) X(17, SceneShell) X(18, SceneSpiral) X(19, SceneTerrain) X(20, SceneTiling) X(21, SceneTree) \
	X(22, SceneDebugMaterials) X(23, SceneNormalTest)
// the first SDFR_PUBLIC_SCENE_COUNT are the reference's scenes (what sdfr_scene_count / sdfr_scene_name list);
// the rest are the library's own diagnostic scenes, loaded by name only (sdfr_scene_debug.h)
enum { SDFR_PUBLIC_SCENE_COUNT = 22, SDFR_SCENE_COUNT = 24 };
// the per-scene kernels are compiled in SDFR_GROUPS translation units (sdfr_kernels_group.hip): scene i in group i % SDFR_GROUPS.
// One scene per unit: the build gives single scenes their own code-generation options (sdf_playground_amd/buildlib.py, SCENE_FLAGS).
#define SDFR_GROUPS 24
#define SDFR_FOR_EACH_GROUP(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23)
static_assert(SDFR_GROUPS >= SDFR_SCENE_COUNT, "one scene per compile unit");

} // namespace sdfr
