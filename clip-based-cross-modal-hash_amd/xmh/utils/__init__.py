"""Small host utilities: config objects, logger, seeding (reference utils/)."""
