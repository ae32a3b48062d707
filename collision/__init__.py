"""Drop-in import path: ``collision.*`` resolves to the MI355X engine in ``collision_amd``."""
