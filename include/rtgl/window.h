// window.h -- headless stand-in for the reference's Window (src/window.h:14-42, src/window.cpp:37-60).
// Same class name, constructor, run() and protected interface; there is no SDL window, GL context or
// ImGui on a compute node, so run() is the bare frame loop:
//
//     m_frames++  ->  tick clock  ->  poll_events()  ->  keyboard_state()  ->  render(dt)  ->  m_time += dt
//
// in exactly the reference's order (m_frames is incremented BEFORE render: the first frame sees 1).
// The loop ends when m_quit is set or after the frame budget (set_frame_budget(), or the environment
// variable RTGL_FRAMES; default 1) -- an interactive window never ends by itself, a batch job must.
// Events can be injected with push_event() to drive the reference's camera / reset handlers.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <string>
#include <vector>

#include "sdl_stub.h"

class Window
{
public:
    struct Clock
    {
        uint64_t last = 0, now = 0;
        float delta = 0.0f;
        static uint64_t ticks() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
        void init() { now = ticks(); }
        void tick() { last = now; now = ticks(); delta = static_cast<float>(now - last) / 1e9f; }
    };

    Window(int width, int height, const std::string &name = "SDL Window") : m_width(width), m_height(height), m_name(name)
    {
        if (const char *e = std::getenv("RTGL_FRAMES")) m_frame_budget = std::atol(e);
        m_keys.assign(SDL_NUM_SCANCODES, 0);
    }
    virtual ~Window() = default;

    void run()
    {
        m_clock.init();
        long done = 0;
        while (!m_quit && (m_frame_budget < 0 || done < m_frame_budget)) {
            m_frames++;
            m_clock.tick();
            poll_events();
            keyboard_state(m_keys.data());
            render(m_clock.delta);
            m_time += m_clock.delta;
            done++;
        }
    }

    // headless extensions (no reference counterpart)
    void set_frame_budget(long frames) { m_frame_budget = frames; }   // < 0: run until m_quit
    void push_event(const SDL_Event &e) { m_events.push_back(e); }
    void set_key(int scancode, bool down) { if (scancode >= 0 && scancode < (int)m_keys.size()) m_keys[scancode] = down ? 1 : 0; }

protected:
    int m_width, m_height;
    int m_frames = 0;
    float m_time = 0.0f;
    bool m_quit = false;
    Clock m_clock;

    void poll_events()
    {
        while (!m_events.empty()) {
            SDL_Event e = m_events.front();
            m_events.pop_front();
            if (e.type == SDL_QUIT) { m_quit = true; break; }
            event(e);
        }
    }

    virtual void render(float dt) { (void)dt; }
    virtual void event(const SDL_Event &event) { (void)event; }
    virtual void keyboard_state(const Uint8 *state) { (void)state; }

private:
    std::string m_name;
    long m_frame_budget = 1;
    std::deque<SDL_Event> m_events;
    std::vector<Uint8> m_keys;
};
