// TEST INFRASTRUCTURE: forward declaration only — nerfstudio.hpp declares functions taking
// nlohmann::json by reference; model.cpp never uses them.
#pragma once
namespace nlohmann { class json; }
