// Ray.h — host-side ray value type (API of reference source/Ray.h:8-28).
// The intersection arithmetic itself runs on the GPU (csrc/rt_device.h); the
// host keeps only the value type user code constructs and passes around.
#pragma once

#include "Vec3.h"

class Ray {
 public:
  Ray(const Vec3f& origin, const Vec3f& direction) : m_origin(origin), m_direction(direction) {}
  const Vec3f& origin() const { return m_origin; }
  const Vec3f& direction() const { return m_direction; }

  // Single ray / single triangle test, Möller–Trumbore with the reference's
  // epsilon and acceptance rules (reference source/Ray.cpp:9-24).  Dispatches one
  // lane of the HIP intersection routine through rt_trace_triangle(); there is no
  // host arithmetic path.  Defined in HostBindings.cpp.
  bool triangleIntersect(const Vec3f& p0, const Vec3f& p1, const Vec3f& p2, float& u, float& v,
                         float& t) const;

 private:
  Vec3f m_origin;
  Vec3f m_direction;
};
