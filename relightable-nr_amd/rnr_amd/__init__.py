"""Host-side layer of the MI355X build: ctypes binding, operators, U-Net plan, pipeline."""
