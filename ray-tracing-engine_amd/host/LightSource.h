// LightSource.h — square area light, host side.
// Constructors / accessors of reference source/LightSource.h:11-66.  The basis
// of the square is built here exactly as the reference's constructor does
// (:28-32); sampling and attenuation run on the GPU.  The reference defines the
// process-wide engine `gen` in this header (:6); the host layer keeps that name
// (one definition, HostBindings.cpp) because user code may seed it.
#pragma once

#include <random>

#include "Vec3.h"

extern std::default_random_engine gen;

class LightSource {
 public:
  LightSource() {}
  LightSource(Vec3f position, Vec3f color, float intensity)
      : m_position(position), m_color(color), m_intensity(intensity) {}
  LightSource(Vec3f position, Vec3f color, Vec3f direction, float intensity, float sideLength)
      : m_position(position), m_color(color), m_direction(direction), m_intensity(intensity),
        m_sideLength(sideLength) {
    m_normal = normalize(m_direction - m_position);
    m_vertical = normalize(cross(m_normal, normalize(m_normal + Vec3f(1.f, 0.f, 0.f))));
    m_horizontal = normalize(cross(m_normal, m_vertical));
  }
  virtual ~LightSource() {}

  Vec3f& position() { return m_position; }
  Vec3f& normal() { return m_normal; }
  Vec3f& color() { return m_color; }
  float intensity() { return m_intensity; }

  const Vec3f& position() const { return m_position; }
  const Vec3f& normal() const { return m_normal; }
  const Vec3f& color() const { return m_color; }
  const Vec3f& vertical() const { return m_vertical; }
  const Vec3f& horizontal() const { return m_horizontal; }
  float intensity() const { return m_intensity; }
  float sideLength() const { return m_sideLength; }
  float factor() const { return m_factor; }
  float attConstant() const { return ac; }
  float attLinear() const { return al; }
  float attQuadratic() const { return aq; }

 private:
  Vec3f m_position, m_color, m_direction, m_normal, m_vertical, m_horizontal;
  float m_factor = 4.5f;
  float m_intensity = 0.f, m_sideLength = 0.f, ac = 1.f, al = 0.3f, aq = 0.3f;
};
