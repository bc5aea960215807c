// Image.h — float RGB frame buffer with the reference's background gradient and
// ASCII PPM writer (reference source/Image.h:10-39, source/Image.cpp:12-43).
#pragma once

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "Vec3.h"

class Image {
 public:
  Image(size_t width = 64, size_t height = 64) : m_width(width), m_height(height) {
    m_pixels.resize(width * height);
  }
  virtual ~Image() {}

  size_t width() const { return m_width; }
  size_t height() const { return m_height; }
  const Vec3f& operator()(size_t x, size_t y) const { return m_pixels[y * m_width + x]; }
  Vec3f& operator()(size_t x, size_t y) { return m_pixels[y * m_width + x]; }

  // raw [h][w][3] float view for the C ABI (Vec3f is three packed floats)
  float* data() { return &m_pixels[0][0]; }
  const float* data() const { return &m_pixels[0][0]; }

  // Vertical gradient from (0.1,0.2,0.8) at the top row to (0.9,0.9,1.0) at the
  // bottom row.  Like the reference (Image.cpp:12-21) the argument is ignored.
  void fillBackground(const Vec3f& /*color*/ = Vec3f(0.f, 0.f, 1.f)) {
    const Vec3f top(0.1f, 0.2f, 0.8f), bottom(0.9f, 0.9f, 1.0f);
    for (size_t y = 0; y < m_height; ++y) {
      float t = static_cast<float>(y) / static_cast<float>(m_height - 1);
      t = t < 0.f ? 0.f : (1.f < t ? 1.f : t);
      const Vec3f c = mix(top, bottom, t);
      for (size_t x = 0; x < m_width; ++x) m_pixels[y * m_width + x] = c;
    }
  }

  // Plain-text P3, one "r g b " triple per pixel, values TRUNCATED (not rounded)
  // from 255*c — byte-compatible with reference Image.cpp:23-43.
  void savePPM(const std::string& filename) const {
    std::ofstream out(filename.c_str());
    if (!out) {
      std::cerr << "Cannot open file " << filename.c_str() << std::endl;
      std::exit(1);
    }
    out << "P3\n" << m_width << " " << m_height << "\n255\n";
    for (const Vec3f& p : m_pixels)
      for (int c = 0; c < 3; ++c) out << static_cast<unsigned int>(255.f * p[c]) << " ";
    out << std::endl;
  }

 private:
  size_t m_width;
  size_t m_height;
  std::vector<Vec3f> m_pixels;
};
