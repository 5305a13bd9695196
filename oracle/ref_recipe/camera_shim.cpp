// oracle/ref_recipe/camera_shim.cpp -- TEST INFRASTRUCTURE.
// A C entry point around the *reference's own* Camera class (compiled from
// /root/reference/Engine/Camera.cpp + Math3D.cpp by oracle/Makefile target `ref`) that
// reproduces how SDFRenderer::render fills the camera constant buffer
// (Engine/SDFRenderer.cpp:85-95) with the set-up calls of Engine/Application.cpp:214-224.
// Output lands in oracle/_ref/ only; no reference source is copied into this repo.
#include "Camera.h"

extern "C" void ref_camera_basis(const float *eye, const float *target, int target_is_direction, float fovy, float aspect, float roll, float *out12)
{
	Camera camera;
	camera.SetCameraMode(Camera::CameraMode::FPS);
	camera.SetAspect(aspect);
	camera.SetFOVY(fovy);
	camera.SetNearPlane(1.f);
	camera.SetFarPlane(300.f);
	camera.SetRoll(roll);
	camera.SetEye(Math3D::Vector3(eye[0], eye[1], eye[2]));
	if (target_is_direction)
		camera.SetDirection(Math3D::Vector3(target[0], target[1], target[2]));
	else
		camera.SetLookat(Math3D::Vector3(target[0], target[1], target[2]));

	Math3D::Vector3 e = camera.GetEye();
	Math3D::Vector3 front = camera.GetDirection();
	Math3D::Vector3 right = (camera.GetFrustrumEdge(0) - camera.GetFrustrumEdge(3)) * 0.5f;
	Math3D::Vector3 top = (camera.GetFrustrumEdge(0) - camera.GetFrustrumEdge(1)) * 0.5f;
	const Math3D::Vector3 v[4] = {e, front, right, top};
	for (int i = 0; i < 4; ++i)
	{
		out12[3 * i + 0] = v[i].x;
		out12[3 * i + 1] = v[i].y;
		out12[3 * i + 2] = v[i].z;
	}
}
