// sdfr_perpixel.h -- the per-pixel pipeline (sdfr_render_pixel.h) together with every scene
// compiled ahead of time, and the registry that numbers them.
#pragma once
#include "sdfr_render_pixel.h"
#include "sdfr_scenes.h"
#include "sdfr_scenes2.h"
#include "sdfr_scenes3.h"
#include "sdfr_scenes4.h"

namespace sdfr {

// scene registry: X(index, SceneType)
#define SDFR_FOR_EACH_SCENE(X) \
	X(0, SceneFastSphere) X(1, SceneCubeSea) X(2, SceneLabyrinth) X(3, SceneFractal) X(4, SceneLense) X(5, SceneGems) X(6, SceneLightShadows) \
	X(7, SceneCube) X(8, SceneGyroid) X(9, SceneBasicTransparency) X(10, SceneBasicClouds) X(11, SceneCoordinateMaterial) \
	X(12, SceneDistortion) X(13, SceneTable) X(14, SceneSierpinski) X(15, SceneNeon) \
	X(16, SceneFractal2) X(17, SceneShell) X(18, SceneSpiral) X(19, SceneTerrain) X(20, SceneTiling) X(21, SceneTree)
enum { SDFR_SCENE_COUNT = 22 };

} // namespace sdfr
