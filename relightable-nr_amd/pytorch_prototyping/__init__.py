"""Parameter containers with the state-dict layout of the reference pytorch_prototyping package."""
