// Scene.h — camera + lights + meshes container (reference source/Scene.h:9-32).
#pragma once

#include <vector>

#include "Camera.h"
#include "LightSource.h"
#include "Mesh.h"

class Scene {
 public:
  Scene() {}
  virtual ~Scene() {}

  const Camera& camera() const { return m_camera; }
  Camera& camera() { return m_camera; }
  const std::vector<LightSource>& lightsources() const { return m_lightsources; }
  std::vector<LightSource>& lightsources() { return m_lightsources; }
  const std::vector<Mesh>& meshes() const { return m_meshes; }
  std::vector<Mesh>& meshes() { return m_meshes; }

 private:
  Camera m_camera;
  std::vector<LightSource> m_lightsources;
  std::vector<Mesh> m_meshes;
};
