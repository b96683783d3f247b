// input.hpp — the headless stand-in for the reference's window input (src/app/window.cppm:68-133): scripted keys and cursor
// travel applied to a scene::Camera, one call per frame.  Used by app::Application (application.hpp) and, for the CPU-side
// known-answer tests, by the host shim (rtr_host_api.cpp).
#pragma once
#include <string>

#include "scene.hpp"

namespace app {

// One frame of Window::processInput + the mouse callback (behaviour of window.cppm:68-133) on scripted input: each of W / S / A / D
// held adds -/+ forward or right times camSpeed to the camera position; T toggles the spin on its press edge, and while the spin is
// on the yaw grows by 0.1 degrees per frame; cursor travel is scaled by mouseSensitivity and handed to Camera::processMouseMovement.
inline void applyInput(scene::Camera& cam, const std::string& keys, float mouseDx, float mouseDy, float camSpeed, float mouseSensitivity,
                       bool& spinning, bool& tWasDown) {
    auto held = [&keys](char upper) { return keys.find(upper) != std::string::npos || keys.find((char)(upper + 32)) != std::string::npos; };
    const bool tDown = held('T');
    if (tDown && !tWasDown) spinning = !spinning;
    tWasDown = tDown;
    // the four movement keys in the reference's order (the sum is rounded after each term, so the order is part of the result)
    rtr::vm::vec3 pos = cam.getPosition();
    const rtr::vm::vec3 fwd = cam.getForward() * camSpeed, right = cam.getRight() * camSpeed;
    bool moved = false;
    if (held('W')) { pos = pos + fwd; moved = true; }
    if (held('S')) { pos = pos - fwd; moved = true; }
    if (held('A')) { pos = pos - right; moved = true; }
    if (held('D')) { pos = pos + right; moved = true; }
    if (moved) cam.setPosition(pos);
    if (mouseDx != 0.0f || mouseDy != 0.0f) cam.processMouseMovement(mouseDx * mouseSensitivity, mouseDy * mouseSensitivity);
    if (spinning) cam.rotateY(0.1f);
}

}  // namespace app
