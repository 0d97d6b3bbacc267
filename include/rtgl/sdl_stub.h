// sdl_stub.h -- the few SDL2 types and constants that appear in the reference's Renderer/Window
// signatures (src/window.h:40-42, src/renderer.cpp:311-420), so code written against that API compiles
// on a compute node without SDL.  Numeric values are SDL2's public ABI values.  If the real SDL2 is on
// the include path, define RTGL_USE_REAL_SDL.
#pragma once
#ifdef RTGL_USE_REAL_SDL
#include <SDL.h>
#else
#include <cstdint>
typedef uint8_t Uint8;
typedef int32_t Sint32;
typedef uint32_t Uint32;
enum { SDL_QUIT = 0x100, SDL_KEYDOWN = 0x300, SDL_KEYUP = 0x301, SDL_MOUSEMOTION = 0x400, SDL_MOUSEBUTTONDOWN = 0x401, SDL_MOUSEBUTTONUP = 0x402 };
enum { SDL_BUTTON_LEFT = 1, SDL_BUTTON_MIDDLE = 2, SDL_BUTTON_RIGHT = 3 };
enum { SDL_SCANCODE_A = 4, SDL_SCANCODE_D = 7, SDL_SCANCODE_E = 8, SDL_SCANCODE_Q = 20, SDL_SCANCODE_S = 22, SDL_SCANCODE_W = 26, SDL_NUM_SCANCODES = 512 };
enum { SDLK_SPACE = ' ', SDLK_j = 'j', SDLK_k = 'k', SDLK_r = 'r' };
struct SDL_Keysym { int scancode; Sint32 sym; uint16_t mod; Uint32 unused; };
struct SDL_KeyboardEvent { Uint32 type, timestamp, windowID; Uint8 state, repeat, padding2, padding3; SDL_Keysym keysym; };
struct SDL_MouseMotionEvent { Uint32 type, timestamp, windowID, which, state; Sint32 x, y, xrel, yrel; };
struct SDL_MouseButtonEvent { Uint32 type, timestamp, windowID, which; Uint8 button, state, clicks, padding1; Sint32 x, y; };
union SDL_Event {
    Uint32 type;
    SDL_KeyboardEvent key;
    SDL_MouseMotionEvent motion;
    SDL_MouseButtonEvent button;
    Uint8 padding[56];
};
#endif
