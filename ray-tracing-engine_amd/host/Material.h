// Material.h — parameters of the microfacet material, host side.
// Constructor/accessor surface of reference source/Material.h:6-23; the BSDF is
// evaluated on the GPU (csrc/rt_device.h: bsdf_eval).  evaluateColorResponse()
// is kept for API compatibility and forwards one evaluation to the device.
#pragma once

#include "Vec3.h"

class Material {
 public:
  Material() : m_kd(3.14159274f), m_alpha(0.5f), m_albedo(0.9f, 0.4f, 0.4f), m_F0(0.31f, 0.31f, 0.31f) {}
  Material(float kd, float alpha, const Vec3f& albedo, const Vec3f& F0)
      : m_kd(kd), m_alpha(alpha), m_albedo(albedo), m_F0(F0) {}
  virtual ~Material() {}

  float kd() const { return m_kd; }
  float alpha() const { return m_alpha; }
  const Vec3f& albedo() const { return m_albedo; }
  const Vec3f& F0() const { return m_F0; }

  // One BSDF evaluation on the device (HostBindings.cpp -> rt_eval_bsdf).
  Vec3f evaluateColorResponse(const Vec3f& normal, const Vec3f& wi, const Vec3f& wo) const;

 private:
  float m_kd, m_alpha;
  Vec3f m_albedo, m_F0;
};
