#include "camera.hpp"

#include <cmath>

// Arithmetic follows the reference (src/camera.cpp:6-57) where it decides the bits:
//   * degrees -> radians goes through double (the reference's PI is a double literal,
//     include/utils.hpp:10,17-19) and is rounded to float; tan/sin/cos are the double routines;
//   * the viewport extents stay double until they scale a vector, where they are rounded to float.
namespace {
constexpr double kPi = 3.14159265359;

float to_radians(float deg) { return (float)(deg * kPi / 180.f); }
}  // namespace

camera::camera(point3 position_, point3 target, float v_fov, float aspect, float aperture, float focus_dist)
    : position(position_), lens_radius(aperture / 2.f), fov(v_fov), focus_distance(focus_dist) {
    forward = (target - position).normalize();
    update_frame();
    set_aspect_ratio(aspect);
}

void camera::update_frame() {
    right = forward.cross(vec3{ 0.0f, 1.0f, 0.0f }).normalize();
    up = right.cross(forward).normalize();
}

void camera::update_viewport() {
    const double half_height = std::tan((double)(to_radians(fov) / 2.f));
    const double viewport_height = 2.f * half_height;
    const double viewport_width = viewport_height * aspect_ratio;

    horizontal = (float)(focus_distance * viewport_width) * right;
    vertical = (float)(focus_distance * viewport_height) * up;
    first_pixel = position - horizontal / 2.0f - vertical / 2.0f + focus_distance * forward;
}

void camera::set_aspect_ratio(float ratio) {
    aspect_ratio = ratio;
    update_viewport();
}

void camera::move(const vec3& v) {
    position += v;
    first_pixel = position - horizontal / 2.0f - vertical / 2.0f + focus_distance * forward;
}

void camera::rotate_y(float theta) {
    // x is updated first and the NEW x feeds z (src/camera.cpp:41-42): kept, it is what the reference does.
    forward.v[0] = (float)(forward.v[0] * std::cos((double)theta) - forward.v[2] * std::sin((double)theta));
    forward.v[2] = (float)(forward.v[0] * std::sin((double)theta) + forward.v[2] * std::cos((double)theta));
    forward.normalize();
    update_frame();
    update_viewport();
}
