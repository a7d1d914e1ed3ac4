"""Stand-ins for the reference's three pybind extension modules (neural_renderer/setup.py:14-27), same module
names, functions backed by the C ABI of librnr_hip.so."""
