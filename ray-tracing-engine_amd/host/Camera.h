// Camera.h — look-at pinhole camera, host side.
// Same constructor arguments and public methods as reference source/Camera.h:7-41;
// adds frame accessors because the GPU path needs the frame (private there).
#pragma once

#include <cmath>

#include "Ray.h"
#include "Vec3.h"

class Camera {
 public:
  Camera(const Vec3f& lookFrom = Vec3f(0.f, 0.f, 1.f), const Vec3f& lookAt = Vec3f(),
         const Vec3f& up = Vec3f(0.f, 1.f, 0.f), float verticalFoV = 60.f, float aspectRatio = 1.f)
      : m_verticalFoV(verticalFoV), m_aspectRatio(aspectRatio), m_position(lookFrom) {
    // Precision notes (they decide the last bit of every primary ray): the
    // degree->radian product is formed in double then narrowed; the tangent is
    // the DOUBLE tangent of the float half-angle, narrowed (Camera.h:14-16 sees
    // only ::tan(double)).
    const float angle = static_cast<float>(static_cast<double>(verticalFoV) * 3.14159265358979323846 /
                                           static_cast<double>(180.f));
    const float halfHeight = static_cast<float>(std::tan(static_cast<double>(angle / 2.f)));
    const float halfWidth = aspectRatio * halfHeight;
    const Vec3f back = normalize(lookFrom - lookAt);
    const Vec3f right = normalize(cross(up, back));
    const Vec3f upv = cross(back, right);
    m_lowerLeftCorner = m_position - halfWidth * right - halfHeight * upv - back;
    m_horizontal = (2.f * halfWidth) * right;
    m_vertical = (2.f * halfHeight) * upv;
  }

  // Primary ray through image-plane coordinate (u, v) in [0,1]^2.  Evaluated by
  // the GPU for every sample (csrc: camera_ray); this host twin exists for API
  // compatibility and uses the same expression order.
  Ray rayAt(float u, float v) const {
    return Ray(m_position, normalize(m_lowerLeftCorner + u * m_horizontal + v * m_vertical - m_position));
  }

  const Vec3f& position() const { return m_position; }
  const Vec3f& lowerLeftCorner() const { return m_lowerLeftCorner; }
  const Vec3f& horizontal() const { return m_horizontal; }
  const Vec3f& vertical() const { return m_vertical; }
  float verticalFoV() const { return m_verticalFoV; }
  float aspectRatio() const { return m_aspectRatio; }

 private:
  float m_verticalFoV;
  float m_aspectRatio;
  Vec3f m_position;
  Vec3f m_lowerLeftCorner;
  Vec3f m_horizontal;
  Vec3f m_vertical;
};
