// Thin-lens look-at camera with the reference's public surface (include/camera.hpp:6-39 in the
// reference).  The object is exactly 128 bytes and is copied verbatim into the first 128 bytes of
// the scene metadata block the kernels read:
//   position@0 forward@16 right@32 up@48 horizontal@64 vertical@80 first_pixel@96
//   lens_radius@112 fov@116 focus_distance@120 aspect_ratio@124
// (The shader declares `up` at 32 and `right` at 48 -- compute.comp:48-50 -- so on the device the
// two are swapped; only the depth-of-field lens offset reads them.  The kernels keep that.)
#ifndef PTR_CAMERA_HPP_
#define PTR_CAMERA_HPP_

#include "vec3.hpp"

class camera {
public:
    camera(point3 position, point3 target, float v_fov, float aspect_ratio, float aperture, float focus_dist);

    void set_aspect_ratio(float ratio);
    void move(const vec3& v);
    void rotate_y(float theta);

    point3 position;

    vec3 forward;
    vec3 right;
    vec3 up;

    vec3 horizontal;
    vec3 vertical;

    vec3 first_pixel;

    float lens_radius;
    float fov;
    float focus_distance;
    float aspect_ratio;

private:
    void update_frame();
    void update_viewport();
};

static_assert(sizeof(camera) == 128, "camera is the first 128 bytes of the scene metadata");

#endif  // PTR_CAMERA_HPP_
