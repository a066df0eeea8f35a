"""Device-side counterparts of the reference's event-representation code."""
