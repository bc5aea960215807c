// Particle.h — one stored photon: position, incoming direction, weight
// (reference source/Particle.h:7-36).
#pragma once

#include <vector>

#include "Vec3.h"

class Particle {
 public:
  Particle() : m_weight(0.f) {}
  Particle(Vec3f position, Vec3f direction, float weight)
      : m_position(position), m_direction(direction), m_weight(weight) {}

  Vec3f& position() { return m_position; }
  const Vec3f& position() const { return m_position; }
  Vec3f& incomeDirection() { return m_direction; }
  const Vec3f& incomeDirection() const { return m_direction; }
  float& weight() { return m_weight; }
  const float& weight() const { return m_weight; }

 private:
  Vec3f m_position, m_direction;
  float m_weight;
};
